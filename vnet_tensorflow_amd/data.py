"""Input side of the hot path: the data contract of the reference's NiftiDataset3D
(pipeline/NiftiDataset3D.py:125-165) without SimpleITK (absent here): image float32
[X,Y,Z,Cin] (axis order after the (2,1,0) transpose, NiftiDataset3D.py:154), label int32
[X,Y,Z] remapped to class *indices* of SegmentationClasses (NiftiDataset3D.py:125-137).

Sources: a case directory tree holding `.npy` or uncompressed NIfTI-1 `.nii` files, or the
synthetic generator of SURVEY.md 8(d) (the shipped sample volumes are Git-LFS stubs).  The
SimpleITK resampling transforms of the YAML pipeline are out of scope;
only the pure index-math RandomCrop to PatchShape is provided."""
import os
import struct

import numpy as np


# ---- minimal NIfTI-1 (single-file, uncompressed) ------------------------------------------------
_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8,
                 512: np.uint16, 768: np.uint32}


def read_nifti(path):
    """Returns (array [X,Y,Z], header dict).  NIfTI stores x fastest, i.e. Fortran order [X,Y,Z]."""
    with open(path, "rb") as f:
        hdr = f.read(348)
        if len(hdr) < 348 or struct.unpack("<i", hdr[:4])[0] != 348:
            raise ValueError("%s: not a little-endian NIfTI-1 file (Git-LFS pointer?)" % path)
        dim = struct.unpack("<8h", hdr[40:56])
        datatype, bitpix = struct.unpack("<hh", hdr[70:74])
        pixdim = struct.unpack("<8f", hdr[76:108])
        vox_offset = struct.unpack("<f", hdr[108:112])[0]
        slope, inter = struct.unpack("<ff", hdr[112:120])
        shape = tuple(int(d) for d in dim[1:1 + dim[0]])
        f.seek(int(vox_offset))
        dt = _NIFTI_DTYPES[datatype]
        data = np.frombuffer(f.read(int(np.prod(shape)) * np.dtype(dt).itemsize), dtype=dt).reshape(shape, order="F")
    if slope not in (0.0, 1.0) or inter != 0.0:
        data = data.astype(np.float32) * (slope if slope != 0 else 1.0) + inter
    return data, {"pixdim": pixdim[1:4], "dim": shape, "datatype": datatype}


def write_nifti(path, arr, pixdim=(1.0, 1.0, 1.0)):
    arr = np.asarray(arr)
    code = {v: k for k, v in _NIFTI_DTYPES.items()}[arr.dtype.type]
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<hh", hdr, 70, code, arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, *pixdim, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<ff", hdr, 112, 1.0, 0.0)
    hdr[344:348] = b"n+1\0"
    with open(path, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0")
        f.write(np.asfortranarray(arr).tobytes(order="F"))


def volume_spacing(path):
    """Voxel spacing (pixdim) of a volume file -- from the 348-byte NIfTI-1 header alone, the voxel array is not read;
    (1, 1, 1) for .npy arrays, which carry none."""
    if path.endswith(".nii"):
        with open(path, "rb") as f:
            hdr = f.read(348)
        if len(hdr) < 348 or struct.unpack("<i", hdr[:4])[0] != 348:
            raise ValueError("%s: not a little-endian NIfTI-1 file (Git-LFS pointer?)" % path)
        return tuple(float(v) for v in struct.unpack("<8f", hdr[76:108])[1:4])
    return (1.0, 1.0, 1.0)


def load_volume(path):
    if path.endswith(".npy"):
        return np.load(path)
    if path.endswith(".nii"):
        return read_nifti(path)[0]
    raise ValueError("unsupported volume format: %s (.npy or uncompressed .nii; SimpleITK is not available)" % path)


def remap_labels(label, classes):
    """NiftiDataset3D.py:125-137: label values -> index into SegmentationClasses (others -> 0)."""
    out = np.zeros(label.shape, dtype=np.int32)
    for idx, c in enumerate(classes):
        out[label == c] = idx
    return out


def synthetic_case(shape, cin, K, seed):
    """One synthetic volume per SURVEY.md 8(d): clamp(127.5 + 40 N(0,1), 0, 255) + 60 inside the
    label spheres; background 0 plus one sphere of radius P/6 per foreground class."""
    rng = np.random.default_rng(seed)
    shape = tuple(shape)
    img = 127.5 + 40.0 * rng.standard_normal(shape + (cin,), dtype=np.float32)
    lab = np.zeros(shape, dtype=np.int32)
    P = min(shape)
    grids = np.ogrid[tuple(slice(0, s) for s in shape)]
    for c in range(1, K):
        ctr = [rng.uniform(s / 4.0, 3.0 * s / 4.0) for s in shape]
        d2 = sum((g - c0) ** 2 for g, c0 in zip(grids, ctr))
        lab[d2 <= (P / 6.0) ** 2] = c
    img += 60.0 * (lab > 0)[..., None]
    return np.clip(img, 0.0, 255.0).astype(np.float32), lab


def random_crop(image, label, patch, rng):
    """Index-math RandomCrop (NiftiDataset3D.py:458-548 without the SimpleITK resampling): pad with
    zeros up to the patch size, then take a uniformly random window."""
    pads = [(0, max(p - s, 0)) for s, p in zip(label.shape, patch)]
    if any(hi for _, hi in pads):
        image = np.pad(image, pads + [(0, 0)])
        label = np.pad(label, pads)
    start = [int(rng.integers(0, s - p + 1)) for s, p in zip(label.shape, patch)]
    sl = tuple(slice(a, a + p) for a, p in zip(start, patch))
    return image[sl], label[sl]


class VolumeDataset(object):
    """Iterates batches (image float32 [B,*P,Cin], label int32 [B,*P,1]) like the reference's
    tf.data pipeline: shuffle, batch(drop_remainder=True) (model.py:289-295).

    Data parallel: every rank shuffles with the SAME generator (seed + epoch), the order is truncated to
    a multiple of world * batch and strided by rank, so the shards are disjoint and every rank yields the same number
    of batches (the gradient all-reduces of the ranks pair up one to one; a rank with fewer steps would leave the
    others blocked in RCCL).  Crop windows are drawn from a per-rank generator.  train=False (the test pass, which holds no
    collective) is not sharded: every rank iterates all cases."""

    def __init__(self, data_dir, image_filenames, label_filename, classes, patch_shape, batch_size,
                 train=True, seed=0, synthetic=None, rank=0, world=1, cache=None, transforms=None):
        self.image_filenames, self.label_filename = list(image_filenames), label_filename
        self.classes, self.patch, self.batch = list(classes), tuple(patch_shape), int(batch_size)
        self.train, self.seed, self.epoch = train, int(seed), 0
        self.rng = np.random.default_rng(seed + 7919 * rank)          # crop windows of this rank
        self.rank, self.world = rank, world
        self.synthetic = synthetic
        # the reference's per-sample transform list (vnet_tensorflow_amd.transforms.build_pipeline of TrainingSetting.Pipeline);
        # None = zero-pad + uniform RandomCrop to PatchShape
        self.transforms = transforms
        # volumes kept in host memory after the first load (synthetic cases are a pure function of their seed;
        # regenerating a 128^3 case costs ~0.3 s, 10x a training step)
        self.cache = {} if (cache if cache is not None else synthetic is not None) else None
        if synthetic is not None:
            self.cases = list(range(int(synthetic.get("Cases", 8))))
        else:
            if not os.path.isdir(data_dir):
                raise FileNotFoundError("data directory %s does not exist" % data_dir)
            self.cases = sorted(os.path.join(data_dir, d) for d in os.listdir(data_dir)
                                if os.path.isdir(os.path.join(data_dir, d)))
        if self.train and self.steps_per_epoch() == 0:
            raise ValueError("%d cases give no full batch for %d rank(s) x batch %d: every rank needs at least one batch per "
                             "epoch (a rank without work would sit in no collective at all)" % (len(self.cases), world, self.batch))

    def steps_per_epoch(self):
        # the TEST pass holds no collective (model.image2label.train: forward + metrics per rank), so it is not sharded: every
        # rank walks every test case, drop_remainder like the reference (model.py:293) -- a test set smaller than
        # world x batch must neither abort the job nor lose cases
        if not self.train:
            return len(self.cases) // self.batch
        return len(self.cases) // (self.world * self.batch)

    def _load(self, case):
        if self.cache is not None and case in self.cache:
            return self.cache[case]
        if self.synthetic is not None:
            shape = self.synthetic.get("Shape", self.patch)
            out = synthetic_case(shape, len(self.image_filenames), len(self.classes),
                                 int(self.synthetic.get("Seed", 1000)) + case)
        else:
            chans = [np.asarray(load_volume(os.path.join(case, f)), dtype=np.float32) for f in self.image_filenames]
            out = np.stack(chans, axis=-1), remap_labels(load_volume(os.path.join(case, self.label_filename)), self.classes)
        if self.cache is not None:
            self.cache[case] = out
        return out

    def epoch_plan(self):
        """[(cases of the batch, crop seeds)] of this rank for the next epoch; advances the epoch counter."""
        order = list(self.cases)
        if self.train:
            np.random.default_rng(self.seed + 104729 * self.epoch).shuffle(order)     # identical on every rank
        self.epoch += 1
        if self.train:
            per = self.world * self.batch
            order = order[:len(order) // per * per][self.rank::self.world]
        else:
            order = order[:len(order) // self.batch * self.batch]
        plan = []
        for i in range(0, len(order), self.batch):
            plan.append((order[i:i + self.batch], [int(v) for v in self.rng.integers(0, 2 ** 62, size=self.batch)]))
        return plan

    def batch_shapes(self):
        """(image shape, label shape) of one batch: [B, *patch, Cin] float32 and [B, *patch, 1] int32."""
        return (self.batch,) + tuple(self.patch) + (len(self.image_filenames),), (self.batch,) + tuple(self.patch) + (1,)

    def make_batch(self, cases, seeds, out=None):
        """out = (image float32 [B,*P,Cin], label int32 [B,*P,1]) NumPy arrays to fill (e.g. views of pinned host tensors: the
        crop is then the ONLY copy between the cached volume and the DMA source); None: new arrays."""
        imgs, labs = [], []
        for i, (case, sd) in enumerate(zip(cases, seeds)):
            image, label = self._load(case)
            if self.transforms is not None:
                from .transforms import apply_pipeline
                image, label = apply_pipeline(self.transforms, image, label, np.random.default_rng(sd))
                if tuple(label.shape) != self.patch:
                    raise ValueError("the transform pipeline produced a %s sample, PatchShape is %s (end it with a crop to "
                                     "PatchShape, like the reference's pipeline3D.yaml)" % (tuple(label.shape), self.patch))
            else:
                image, label = random_crop(image, label, self.patch, np.random.default_rng(sd))
            if out is not None:
                np.copyto(out[0][i], image, casting="unsafe")
                np.copyto(out[1][i, ..., 0], label, casting="unsafe")
                continue
            imgs.append(image)
            labs.append(label[..., None])
        if out is not None:
            return out
        return np.stack(imgs).astype(np.float32, copy=False), np.stack(labs).astype(np.int32, copy=False)

    def __iter__(self):
        for cases, seeds in self.epoch_plan():
            yield self.make_batch(cases, seeds)


class Prefetcher(object):
    """Runs `dataset.make_batch` on worker threads (NumPy releases the GIL in the copies that dominate it), `depth`
    batches ahead of the consumer and in order, and hands them over as PINNED host tensors, so that the training loop's
    host-to-device copy is asynchronous (`non_blocking=True`) and overlaps the previous step's kernels.
    The counterpart of the reference's tf.data `prefetch` (model.py:289-295)."""

    def __init__(self, dataset, depth=4, workers=3, pin=True):
        self.dataset, self.depth, self.workers, self.pin = dataset, int(depth), int(workers), pin

    def _job(self, cases, seeds):
        import torch
        if self.pin and torch.cuda.is_available() and len(cases) == self.dataset.batch and hasattr(self.dataset, "batch_shapes"):
            # crop straight into pinned memory (torch's caching host allocator hands a block out again only after the
            # asynchronous copies that read it have finished): one host copy per batch instead of three
            si, sl = self.dataset.batch_shapes()
            ti = torch.empty(si, dtype=torch.float32, pin_memory=True)
            tl = torch.empty(sl, dtype=torch.int32, pin_memory=True)
            self.dataset.make_batch(cases, seeds, out=(ti.numpy(), tl.numpy()))
            return ti, tl
        img, lab = self.dataset.make_batch(cases, seeds)
        ti, tl = torch.from_numpy(img), torch.from_numpy(lab)
        if self.pin and torch.cuda.is_available():
            ti, tl = ti.pin_memory(), tl.pin_memory()
        return ti, tl

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        plan = self.dataset.epoch_plan()
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            pending, it = deque(), iter(plan)
            for cases, seeds in it:
                pending.append(pool.submit(self._job, cases, seeds))
                if len(pending) >= self.depth:
                    break
            while pending:
                out = pending.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(self._job, *nxt))
                yield out

"""Array restatements of the reference's 3-D sample transforms that are pure index / intensity arithmetic
(pipeline/NiftiDataset3D.py) -- the input side of the hot path without SimpleITK (absent in this image):

    StatisticalNormalization  NiftiDataset3D.py:210-254      ManualNormalization  NiftiDataset3D.py:285-308
    ExtremumNormalization     NiftiDataset3D.py:256-283      Normalization        NiftiDataset3D.py:167-185
    RandomFlip                NiftiDataset3D.py:187-208      Padding              NiftiDataset3D.py:400-456
    RandomCrop                NiftiDataset3D.py:458-551      RandomNoise          NiftiDataset3D.py:553-572
    ConfidenceCrop2           NiftiDataset3D.py:661-793

A sample is {'image': float32 [X,Y,Z,C] (the reference keeps a list of C SimpleITK images), 'label': int [X,Y,Z]}; every
transform is `t(sample, rng)` with an explicit numpy Generator (the reference draws from the global `random` / `np.random`
state).  `build_pipeline` reads the reference's YAML schema (pipeline/pipeline3D.yaml: preprocess -> train|test|evaluate ->
3D -> [{name, variables}], model.py:340-372) and instantiates by class name exactly like model.py:350.  Transforms that
RESAMPLE on a physical grid (Resample, Reorient, Invert, BSplineDeformation, ConfidenceCrop) need SimpleITK's geometry and
are out of scope: naming one raises."""
import numpy as np

_SITK_ONLY = ("Resample", "Reorient", "Invert", "BSplineDeformation", "ConfidenceCrop")


def _size3(v, what):
    if isinstance(v, int):
        return (v, v, v)
    assert isinstance(v, (tuple, list)) and len(v) == 3, "%s: int or 3 values" % what
    return tuple(int(x) for x in v)


def _window(img, wmin, wmax, omin=0.0, omax=255.0):
    """sitk.IntensityWindowingImageFilter: linear map of [wmin, wmax] onto [omin, omax], values outside clamp."""
    x = np.asarray(img, dtype=np.float64)
    if wmax == wmin:
        return np.where(x < wmin, omin, omax).astype(np.float32)
    y = (x - wmin) * ((omax - omin) / (wmax - wmin)) + omin
    return np.clip(y, omin, omax).astype(np.float32)


class Normalization(object):
    """sitk.RescaleIntensityImageFilter to [0, 255] (per channel)."""
    name = 'Normalization'

    def __call__(self, sample, rng=None):
        img = sample['image']
        out = np.empty(img.shape, dtype=np.float32)
        for c in range(img.shape[-1]):
            ch = img[..., c].astype(np.float64)
            out[..., c] = _window(ch, ch.min(), ch.max())
        return {'image': out, 'label': sample['label']}


class StatisticalNormalization(object):
    """Window [mean - sigma*std, mean + sigma*std] -> [0, 255] per channel; std is ITK's (unbiased, N-1)."""

    def __init__(self, sigma, pre_norm=False):
        self.name = 'StatisticalNormalization'
        assert isinstance(sigma, float)
        self.sigma, self.pre_norm = sigma, pre_norm

    def __call__(self, sample, rng=None):
        img = sample['image']
        out = np.empty(img.shape, dtype=np.float32)
        for c in range(img.shape[-1]):
            ch = img[..., c].astype(np.float64)
            if self.pre_norm:                                   # sitk.NormalizeImageFilter: zero mean, unit variance
                ch = (ch - ch.mean()) / ch.std(ddof=1)
            mu, sd = ch.mean(), ch.std(ddof=1)
            fi = np.finfo(np.float32)
            wmax = min(mu + self.sigma * sd, float(fi.max))
            wmin = max(mu - self.sigma * sd, float(fi.min))
            out[..., c] = _window(ch, wmin, wmax)
        return {'image': out, 'label': sample['label']}


class ExtremumNormalization(object):
    def __init__(self, percent=0.05):
        self.name = 'ExtremumNormalization'
        assert isinstance(percent, float)
        self.percent = percent

    def __call__(self, sample, rng=None):
        img = sample['image']
        out = np.empty(img.shape, dtype=np.float32)
        for c in range(img.shape[-1]):
            ch = img[..., c].astype(np.float64)
            lo, hi = ch.min(), ch.max()
            out[..., c] = _window(ch, (hi - lo) * self.percent + lo, (hi - lo) * (1 - self.percent) + lo)
        return {'image': out, 'label': sample['label']}


class ManualNormalization(object):
    def __init__(self, windowMin, windowMax):
        self.name = 'ManualNormalization'
        assert isinstance(windowMax, (int, float)) and isinstance(windowMin, (int, float))
        self.windowMax, self.windowMin = float(windowMax), float(windowMin)

    def __call__(self, sample, rng=None):
        return {'image': _window(sample['image'], self.windowMin, self.windowMax), 'label': sample['label']}


class RandomFlip(object):
    """One coin flip per sample; heads flips image and label along every axis whose entry in `axes` is true."""

    def __init__(self, axes):
        self.name = 'Flip'
        assert len(axes) > 0 and len(axes) <= 3
        self.axes = axes

    def __call__(self, sample, rng):
        image, label = sample['image'], sample['label']
        if int(rng.integers(2)):
            ax = tuple(i for i, f in enumerate(self.axes) if f)
            image, label = np.flip(image, ax), np.flip(label, ax)
        return {'image': np.ascontiguousarray(image), 'label': np.ascontiguousarray(label)}


class Padding(object):
    """Grow the volume to at least output_size: the reference resamples onto a larger grid with the same origin, spacing
    and direction, i.e. the old voxels keep their indices and the new ones (high side of each axis) are 0."""

    def __init__(self, output_size):
        self.name = 'Padding'
        self.output_size = _size3(output_size, 'output_size')
        assert all(i > 0 for i in self.output_size)

    def __call__(self, sample, rng=None):
        image, label = sample['image'], sample['label']
        old = label.shape
        if all(o >= n for o, n in zip(old, self.output_size)):
            return sample
        pads = [(0, max(n - o, 0)) for o, n in zip(old, self.output_size)]
        return {'image': np.pad(image, pads + [(0, 0)]), 'label': np.pad(label, pads)}


class RandomCrop(object):
    """Random window of output_size; a window with fewer than min_pixel foreground voxels is kept only with probability
    drop_ratio, otherwise another one is drawn (NiftiDataset3D.py:518-541).  np.random.randint(0, n) of the reference
    excludes n, so the last admissible start index (old - new) is never drawn -- kept."""

    def __init__(self, output_size, drop_ratio=0.1, min_pixel=1):
        self.name = 'Random Crop'
        self.output_size = _size3(output_size, 'output_size')
        assert isinstance(drop_ratio, (int, float))
        if not 0 <= drop_ratio <= 1:
            raise RuntimeError('Drop ratio should be between 0 and 1')
        assert isinstance(min_pixel, int)
        if min_pixel < 0:
            raise RuntimeError('Min label pixel count should be integer larger than 0')
        self.drop_ratio, self.min_pixel = drop_ratio, min_pixel

    def __call__(self, sample, rng):
        image, label = sample['image'], sample['label']
        old, new = label.shape, self.output_size
        fg = (label >= 1) & (label <= 255)
        while True:
            start = [0 if o <= n else int(rng.integers(0, o - n)) for o, n in zip(old, new)]
            sl = tuple(slice(s, s + n) for s, n in zip(start, new))
            if int(fg[sl].sum()) >= self.min_pixel or rng.random() <= self.drop_ratio:
                break
        return {'image': np.ascontiguousarray(image[sl]), 'label': np.ascontiguousarray(label[sl])}


class RandomNoise(object):
    """sitk.AdditiveGaussianNoiseImageFilter(mean 0, standard deviation sigma) on every channel."""

    def __init__(self, sigma=5):
        self.name = 'Random Noise'
        self.sigma = sigma

    def __call__(self, sample, rng):
        image = sample['image']
        noise = rng.standard_normal(image.shape, dtype=np.float32) * np.float32(self.sigma)
        return {'image': image + noise, 'label': sample['label']}


class ConfidenceCrop2(object):
    """With probability `probability` (in tenths, as the reference's choice list) crop around the bounding-box centre of a
    randomly chosen connected label component, offset by a uniform integer in [-rand_range, rand_range] per axis; otherwise
    (or when there is no label) a random region -- optionally one without any label (NiftiDataset3D.py:661-793)."""

    def __init__(self, output_size, rand_range=3, probability=0.5, random_empty_region=False):
        self.name = 'Confidence Crop 2'
        self.output_size = _size3(output_size, 'output_size')
        self.rand_range = _size3(rand_range, 'rand_range')
        assert isinstance(probability, float) and 0 <= probability <= 1
        self.probability = probability
        assert isinstance(random_empty_region, bool)
        self.random_empty_region = random_empty_region

    def _crop(self, image, label, index):
        sl = tuple(slice(i, i + n) for i, n in zip(index, self.output_size))
        return np.ascontiguousarray(image[sl]), np.ascontiguousarray(label[sl])

    def _random_index(self, size, rng):
        # random.choice(range(0, size - out - 1)): the last two admissible starts are never drawn (reference as written)
        idx = []
        for s, n in zip(size, self.output_size):
            if s - n == 0:
                idx.append(0)
            else:
                hi = s - n - 1
                if hi <= 0:
                    raise IndexError("Cannot choose from an empty sequence")          # what random.choice(range(0, 0)) raises
                idx.append(int(rng.integers(0, hi)))
        return idx

    def RandomRegion(self, image, label, rng):
        return self._crop(image, label, self._random_index(label.shape, rng))

    def RandomEmptyRegion(self, image, label, rng):
        while True:
            img, lab = self._crop(image, label, self._random_index(label.shape, rng))
            if lab.sum() < 1:
                return img, lab

    def __call__(self, sample, rng):
        from scipy import ndimage
        image, label = sample['image'], sample['label'].astype(np.int16)
        choices = [0] * int(10 * (1 - self.probability)) + [1] * int(10 * self.probability)
        label_type = choices[int(rng.integers(len(choices)))]
        pick_random = self.RandomEmptyRegion if self.random_empty_region else self.RandomRegion
        if label_type == 0:
            image, label = pick_random(image, label, rng)
            return {'image': image, 'label': label}
        cc, n = ndimage.label(label != 0)           # face connectivity = sitk.ConnectedComponentImageFilter default
        if n == 0:
            image, label = pick_random(image, label, rng)
            return {'image': image, 'label': label}
        sel = int(rng.integers(n)) + 1
        box = ndimage.find_objects((cc == sel).astype(np.uint8))[0]
        index = []
        for i in range(3):
            lo, ext = box[i].start, box[i].stop - box[i].start
            ix = lo + int(ext / 2) - int(self.output_size[i] / 2) + int(rng.integers(-self.rand_range[i], self.rand_range[i] + 1))
            if label.shape[i] - ix - 1 < self.output_size[i]:
                ix = label.shape[i] - self.output_size[i] - 1
            if ix < 0:
                ix = 0
            index.append(ix)
        image, label = self._crop(image, label, index)
        return {'image': image, 'label': label}


_REGISTRY = {c.__name__: c for c in (Normalization, StatisticalNormalization, ExtremumNormalization, ManualNormalization,
                                     RandomFlip, Padding, RandomCrop, RandomNoise, ConfidenceCrop2)}


def build_pipeline(yaml_path, phase):
    """[transform] of preprocess -> `phase` ('train' | 'test' | 'evaluate') -> 3D of a reference pipeline YAML."""
    import yaml
    with open(yaml_path) as f:
        spec = yaml.load(f, Loader=yaml.SafeLoader)
    entries = (spec.get("preprocess", {}).get(phase, {}) or {}).get("3D") or []
    out = []
    for t in entries:
        name = t["name"]
        if name in _SITK_ONLY:
            raise NotImplementedError("transform %r resamples on the physical grid and needs SimpleITK (out of scope here); "
                                      "resample the volumes offline and drop it from the pipeline" % name)
        if name not in _REGISTRY:
            raise AttributeError("module 'NiftiDataset3D' has no attribute %r" % name)
        out.append(_REGISTRY[name](**(t.get("variables") or {})))
    return out


def apply_pipeline(transforms, image, label, rng):
    sample = {'image': image, 'label': label}
    for t in transforms:
        sample = t(sample, rng)
    return sample['image'], sample['label']

"""Mirror of the reference's model.py for the V-Net hot path: `dice_coe` (model.py:26-85) and
`class image2label` (model.py:169-1242) with `.train()` / `.evaluate()`, driven by the same
config.json contract (model.py:185-245) -- on the HIP library instead of a tf.Session.

What is carried over: the graph of model.py:428-568 (network, softmax, one-hot, the Loss.Name
switch, argmax), the optimiser/LR schedule of model.py:641-666, the step loop of model.py:716-810
(feed images/labels, dropout from config, batch-statistics BN, one optimiser step, print loss,
checkpoint cadence, periodic test batch) and the sliding-window inference of model.py:866-937.
What is not: SimpleITK I/O + resampling transforms, TensorBoard summaries,
the 2-D path, UNet.  `sess` is accepted and ignored.
"""
import datetime
import math
import os
import shutil
import sys
import time

import numpy as np
import torch

from . import data as vdata
from . import networks, ops, optim, parallel
from ._lib import VnetHipError, check, lib


# ------------------------------------------------------------------------------------------------
# dice_coe -- reference model.py:26-85 (same signature and defaults)
# ------------------------------------------------------------------------------------------------
class _DiceCoeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, jaccard, weights, smooth):
        L = lib()
        output, target = output.contiguous(), target.contiguous()
        B, K = output.shape[0], output.shape[-1]
        V = output.numel() // (B * K)
        dev = output.device
        dice = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(2 * B * K + 1, dtype=torch.float32, device=dev)
        nb = L.vnet_loss_ws_bytes(B, K) + 4
        ws = ops.workspace(nb, dev)
        check(L.vnet_dice_coe_fwd(ops._ptr(output), ops._ptr(target), B, V, K, int(jaccard), ops._ptr(weights), smooth,
                                  ops._ptr(dice), ops._ptr(coef), ops._ptr(ws), nb, ops._stream()), "vnet_dice_coe_fwd")
        ctx.save_for_backward(output, target, coef)
        ctx.cfg = (B, V, K, int(jaccard))
        return dice

    @staticmethod
    def backward(ctx, g):
        output, target, coef = ctx.saved_tensors
        B, V, K, jac = ctx.cfg
        d = torch.empty_like(output)
        check(lib().vnet_dice_coe_bwd(ops._ptr(output), ops._ptr(target), B, V, K, jac, ops._ptr(coef),
                                      ops._ptr(g.contiguous()), ops._ptr(d), ops._stream()), "vnet_dice_coe_bwd")
        return d, None, None, None, None


def dice_coe(output, target, loss_type='jaccard', axis=(1, 2, 3), weights=[], smooth=1e-5):
    """Soft dice (Sorensen or Jaccard) coefficient, reference model.py:26-85.  `output` / `target`:
    float32 [B, D, H, W, K] on the HIP device; reduction over `axis` = all spatial axes."""
    if loss_type not in ('jaccard', 'sorensen'):
        raise Exception("Unknown loss_type")
    if tuple(axis) != tuple(range(1, output.dim() - 1)):
        raise NotImplementedError("dice_coe reduces over all spatial axes (the only use in the reference)")
    ops._need_gpu(output, "dice_coe")
    wt = None
    if len(weights) != 0:
        assert len(weights) == target.shape[-1], "Length of DICE weight is {}, should be {}".format(len(weights), target.shape[-1])
        wt = torch.as_tensor(list(weights), dtype=torch.float32).to(output.device)
    return _DiceCoeFn.apply(output, target.to(torch.float32), loss_type == 'jaccard', wt, float(smooth))


def volume_threshold(label, volume, spacing=(1.0, 1.0, 1.0)):
    """reference model.py:117-140: keep (as 1) every face-connected component of the non-zero label whose PHYSICAL size
    (voxels x voxel volume) exceeds `volume`; the result is a uint8 0/1 map."""
    from scipy import ndimage
    cc, n = ndimage.label(np.asarray(label) != 0)
    if n == 0:
        return np.zeros(np.shape(label), dtype=np.uint8)
    sizes = np.bincount(cc.ravel(), minlength=n + 1).astype(np.float64) * float(np.prod(spacing))
    keep = sizes > volume
    keep[0] = False
    return keep[cc].astype(np.uint8)


def ExtractLargestConnectedComponents(label, spacing=(1.0, 1.0, 1.0)):
    """reference model.py:142-167: 1 on the largest face-connected component of the (uint8-cast) non-zero label, 0 elsewhere
    (first one wins a tie, like the reference's strict `>` scan in label order)."""
    from scipy import ndimage
    cc, n = ndimage.label(np.asarray(label).astype(np.uint8) != 0)
    if n == 0:
        return np.zeros(np.shape(label), dtype=np.uint8)
    sizes = np.bincount(cc.ravel(), minlength=n + 1)
    sizes[0] = 0
    return (cc == int(np.argmax(sizes))).astype(np.uint8)


def prepare_batch(image_ijk_patch_indices_dict):
    """reference model.py:94-115"""
    images, ijk = image_ijk_patch_indices_dict['images'], image_ijk_patch_indices_dict['indexes']
    return np.asarray([images[p[0]:p[1], p[2]:p[3], p[4]:p[5], :] for p in ijk])


def export_weights(network, path):
    """Write every trainable variable of `network` (TF names) as the blob the native driver loads
    (csrc/vnet_infer.cpp): 'VNETW1\\0\\0', u32 nvars, then {u32 name_len, name, u32 ndim, u32 dims[], f32 data}.
    The counterpart of the reference's meta_to_pb.py (checkpoint -> graph.pb for cxx/)."""
    import struct
    items = network.named_parameters()
    with open(path, "wb") as f:
        f.write(b"VNETW1\0\0")
        f.write(struct.pack("<I", len(items)))
        for name, p in items:
            arr = np.ascontiguousarray(p.detach().cpu().numpy(), dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb)
            f.write(struct.pack("<I", arr.ndim) + struct.pack("<%dI" % arr.ndim, *arr.shape))
            f.write(arr.tobytes())


def _now():
    return datetime.datetime.now()


class _DeviceFeeder(object):
    """Double-buffered host-to-device input path of the training loop: batch t+1 is copied from pinned host memory into
    its device slot on a COPY stream while step t runs; the compute stream only waits for the copy's event.  A slot is
    rewritten only after the step that read it has finished (event recorded behind that step).  With this the
    PCIe transfer (16.8 MB per 128^3 fp32 patch + int32 labels) is off the step's critical path."""

    def __init__(self, device, loader):
        self.device, self.it = device, iter(loader)
        self.cuda = device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.slots = [None, None]
        self.ready = [None, None]
        self.done = [None, None]
        self.t = 0
        self._stage(0)

    def _stage(self, k):
        try:
            img, lab = next(self.it)
        except StopIteration:
            self.slots[k] = None
            return
        img = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
        lab = lab if isinstance(lab, torch.Tensor) else torch.from_numpy(lab)
        if not self.cuda:
            self.slots[k] = (img.to(self.device), lab.to(self.device))
            return
        with torch.cuda.stream(self.copy_stream):
            if self.done[k] is not None:
                self.copy_stream.wait_event(self.done[k])          # the step that read this slot has finished
            old = self.slots[k]
            if old is not None and old[0].shape == img.shape and old[1].shape == lab.shape:
                di, dl = old
                di.copy_(img, non_blocking=True)
                dl.copy_(lab, non_blocking=True)
            else:
                di, dl = img.to(self.device, non_blocking=True), lab.to(self.device, non_blocking=True)
            self.slots[k] = (di, dl)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self.ready[k] = ev

    def __iter__(self):
        return self

    def __next__(self):
        k = self.t % 2
        cur = self.slots[k]
        if cur is None:
            raise StopIteration
        self._stage(1 - k)                                          # batch t+1 starts its transfer now
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        self._cur = k
        self.t += 1
        return cur

    def step_done(self):
        """Call after the step that consumed the last batch has been enqueued."""
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.done[self._cur] = ev


# ------------------------------------------------------------------------------------------------
# image2label -- reference model.py:169-1242
# ------------------------------------------------------------------------------------------------
def in_context(fn):
    """The method runs inside the model's own OpsContext (compute dtype, parameter-gradient stream, packed-filter registry): what
    used to be process-wide switches is per-model state, so models of different ComputeDtype coexist and interleave their steps."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with ops.context(self.ctx):
            return fn(self, *a, **kw)
    return wrapped


class image2label(object):
    def __init__(self, sess, config, device=None, verbose=True):
        self.ctx = ops.OpsContext()
        """Args: sess: ignored (kept for signature parity, model.py:170); config: parsed config.json"""
        self.sess = sess
        self.config = config
        self.model = None
        self.epoches = 999999999999999999
        self.verbose = verbose
        self.rank, self.local_rank, self.world = 0, 0, 1
        self.device = torch.device(device) if device is not None else None
        self.network = None
        self.global_step = 0
        self.start_epoch = 0
        self.momentum = 0.9
        self.last_loss = None

    def _print(self, *a):
        if self.verbose and self.rank == 0:
            print(*a)

    def _device_sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    # -- reference model.py:185-245, with the corrected key set of SURVEY.md B.3 --------------------
    def read_config(self):
        self._print("{}: Reading configuration file...".format(_now()))
        T = self.config['TrainingSetting']
        self.input_channel_num = len(T['Data']['ImageFilenames'])
        self.output_channel_num = len(T['SegmentationClasses'])
        self.label_classes = T['SegmentationClasses']
        self.train_data_dir = T['Data']['TrainingDataDirectory']
        self.test_data_dir = T['Data']['TestingDataDirectory']
        self.image_filenames = T['Data']['ImageFilenames']
        self.label_filename = T['Data']['LabelFilename']
        self.synthetic = T['Data'].get('Synthetic')            # extension: synthetic generator (no NIfTI shipped)
        self.compute_dtype = T.get('ComputeDtype', 'fp32')            # extension: 'bf16' = BASELINE config C5 arithmetic
        # 'fp32_split3' (round 5): fp32 tensors and fp32 accuracy, the 5^3 convolutions on the bf16 matrix pipe (csrc/conv_x3.h)
        if self.compute_dtype not in ('fp32', 'fp32_split3', 'bf16'):
            raise SystemExit("Invalid ComputeDtype %r (fp32 | fp32_split3 | bf16; round 2's 'bf16_operands' was retired in round 5: "
                             "'bf16' is its successor)" % (self.compute_dtype,))
        if self.compute_dtype == 'bf16':
            # 'bf16' = bf16 STORAGE (every activation a bf16 tensor); its kernels move 8-channel units whose count is a power of
            # two: fail HERE, not at the first forward
            nch = int(T.get('Networks', {}).get('NumChannel', 16))
            if nch < 8 or nch & (nch - 1):
                raise SystemExit("ComputeDtype 'bf16' (bf16 storage) needs Networks.NumChannel = 8 * 2^k, got %d; "
                                 "'fp32' / 'fp32_split3' take any width" % nch)
        self.sync_batch_norm = bool(T.get('SyncBatchNorm', False))   # extension: cross-replica BN statistics
        # extension (round 6, opt-in): gradient buckets travel as bf16 -- half the bytes over xGMI -- and are accumulated in fp32
        # on receipt (parallel.BucketedGradAllReduce); only with ComputeDtype 'bf16', whose gradients come from bf16 tensors anyway
        self.grad_comm_dtype = T.get('GradCommDtype', 'fp32')
        if self.grad_comm_dtype not in ('fp32', 'bf16'):
            raise SystemExit("Invalid GradCommDtype %r (fp32 | bf16)" % (self.grad_comm_dtype,))
        if self.grad_comm_dtype == 'bf16' and self.compute_dtype != 'bf16':
            raise SystemExit("GradCommDtype 'bf16' needs ComputeDtype 'bf16' (the fp32 modes keep fp32 gradients on the links)")
        if 'AllReduceHoldFraction' in T:                               # extension: when the gradient buckets are launched (parallel.py)
            self.allreduce_hold_fraction = float(T['AllReduceHoldFraction'])
        if 'PrefetchDepth' in T:                                       # extension: input prefetcher of train() (0 = synchronous loader)
            self.prefetch_depth = int(T['PrefetchDepth'])
        if 'LoaderThreads' in T:
            self.loader_threads = int(T['LoaderThreads'])
        self.batch_size = T['BatchSize']
        self.patch_shape = T['PatchShape']
        self.dimension = len(T['PatchShape'])
        self.image_log = T.get('ImageLog', False)
        self.testing = T.get('Testing', False)
        self.test_step = T.get('TestStep', 100)                 # missing from the shipped config.json
        self.restore_training = T.get('Restore', False)
        self.log_dir = T.get('LogDir', './tmp/log')
        self.ckpt_dir = T.get('CheckpointDir', './tmp/ckpt')
        self.epoches = T.get('Epoches', 1)
        self.max_itr = T.get('MaxIterations', 10 ** 12)        # missing from the shipped config.json
        self.log_interval = T.get('LogInterval', 100)
        N = T['Networks']
        self.network_name = N['Name']
        self.dropout_rate = N['Dropout']
        self.num_channel = N['NumChannel']
        self.num_levels = N['NumLevels']
        # the shipped JSONs spell it "NumCovolutions" (configs/config.json:29); accept both
        self.num_convolutions = N['NumConvolutions'] if 'NumConvolutions' in N else N['NumCovolutions']
        self.bottom_convolutions = N['BottomConvolutions']
        O = T['Optimizer']
        self.optimizer_name = O['Name']
        self.initial_learning_rate = O['InitialLearningRate']
        self.decay_factor = O['Decay']['Factor']
        self.decay_steps = O['Decay']['Steps']
        self.momentum = O.get('Momentum', 0.9)                  # the reference never sets self.momentum (model.py:654)
        self.spacing = T.get('Spacing')
        self.drop_ratio = T.get('DropRatio')
        self.min_pixel = T.get('MinPixel')
        self.loss_name = T['Loss']['Name']
        self.loss_weights = T['Loss'].get('Weights', [])
        self.loss_alpha = T['Loss'].get('Alpha', 1)
        self.training_pipeline = T.get('Pipeline')
        E = self.config.get('EvaluationSetting', {})
        self.checkpoint_path = E.get('CheckpointPath')
        ED = E.get('Data', {})
        self.evaluate_data_dir = ED.get('EvaluateDataDirectory')
        self.evaluate_image_filenames = ED.get('ImageFilenames', self.image_filenames)
        self.evaluate_label_filename = ED.get('LabelFilename', 'label_tf.nii')
        self.evaluate_probability_filename = ED.get('ProbabilityFilename', 'probability_tf.nii')
        self.evaluate_stride = E.get('Stride', self.patch_shape)
        self.evaluate_batch = E.get('BatchSize', 1)
        self.evaluate_probability_output = E.get('ProbabilityOutput', False)
        self.evaluate_lcc = E.get('LargestConnectedComponent', False)
        self.evaluate_volume_threshold = E.get('VolumeThreshold', False)
        self.evaluate_pipeline = E.get('Pipeline')
        self._print("{}: Reading configuration file complete".format(_now()))

    # -- reference model.py:297-630 (network + loss head; summaries/metrics not carried) ---------------
    @in_context
    def build_model_graph(self):
        self._print("{}: Start to build model graph...".format(_now()))
        self._validate_loss()
        if self.dimension != 3:
            sys.exit("Only 3D PatchShape is built (2D is out of scope)")
        if self.device is None:
            if not torch.cuda.is_available():
                raise VnetHipError("no HIP device: the MI355X kernels are the only compute path")
            self.device = torch.device("cuda", self.local_rank)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)     # kernels are launched on the CURRENT device's current stream (ops._raw_stream)
        self.input_batch_shape = (self.batch_size,) + tuple(self.patch_shape) + (self.input_channel_num,)
        self.output_batch_shape = (self.batch_size,) + tuple(self.patch_shape) + (1,)
        self.dropout_placeholder = self.dropout_rate     # stand-in for "dropout_placeholder" (model.py:312)
        ops.set_compute_dtype(getattr(self, "compute_dtype", "fp32"))
        if self.network_name == "VNet":
            self.network = networks.VNet(
                num_classes=self.output_channel_num,
                dropout_rate=lambda: self.dropout_placeholder,
                num_channels=self.num_channel,
                num_levels=self.num_levels,
                num_convolutions=self.num_convolutions,
                bottom_convolutions=self.bottom_convolutions,
                is_training=True,
                activation_fn="prelu",
                device=self.device)
        else:
            sys.exit("Invalid Network")
        self.network.build(self.input_batch_shape)
        self._print("{}: Core network complete".format(_now()))

    def _validate_loss(self):
        """Loss.Name check of model.py:495-560, INCLUDING its fall-through: `if name == "xent"` (model.py:495) is a
        separate statement from the `if name == "weighted_xent" / elif ... / else: sys.exit("Invalid loss function")` chain
        that follows (model.py:497-560), so the reference builds the xent loss and then exits.  A drop-in keeps that
        (nobody can be training with it); `Loss.AllowPlainXent: true` (extension) runs the plain cross-entropy instead."""
        ops.parse_loss(self.loss_name)                   # unknown names: SystemExit("Invalid loss function")
        if self.loss_name == "xent" and not self.config['TrainingSetting']['Loss'].get('AllowPlainXent', False):
            sys.exit("Invalid loss function")

    @in_context
    def forward(self, images, labels=None, dropout=0.0, want_softmax=False, want_pred=False):
        """One pass of the graph of model.py:444-568.  images float32 [B,*P,Cin]; labels int32 [B,*P,1]."""
        self.dropout_placeholder = dropout
        logits = self.network.GetNetwork(images)
        if labels is None:
            sm, pred = ops.softmax_argmax(logits)
            return logits, None, sm, pred
        loss, dice, sm, pred = ops.softmax_loss(logits, labels, self.loss_name, self.loss_weights, self.loss_alpha,
                                                want_softmax=want_softmax, want_pred=want_pred)
        return logits, loss, sm, pred

    @in_context
    def run(self, fetches, feed_dict):
        """sess.run shim keyed by the reference's graph tensor names (model.py:914-917)."""
        img = feed_dict['images_placeholder:0']
        img = torch.as_tensor(np.ascontiguousarray(img), dtype=torch.float32).to(self.device)
        lab = feed_dict.get('labels_placeholder:0')
        if lab is not None:
            lab = torch.as_tensor(np.ascontiguousarray(lab), dtype=torch.int32).to(self.device)
        with torch.no_grad():
            logits, loss, sm, pred = self.forward(img, lab, float(feed_dict.get('dropout_placeholder:0', 0.0)), True, True)
        table = {'softmax:0': sm, 'predicted_label/prediction:0': pred, 'logits:0': logits, 'loss:0': loss}
        return [table[f].cpu().numpy() for f in fetches]

    # -- distributed / optimiser set-up -------------------------------------------------------------
    @in_context
    def _setup_training(self):
        self.flat = optim.FlatParams(self.network.named_parameters())
        self.optimizer = optim.make_optimizer(self.optimizer_name, self.flat, self.momentum)
        self.sync = None
        self._two_pass = False
        pg = os.environ.get("VNET_PARAM_GRAD_STREAM")
        ops.set_param_grad_stream(self.device.type == "cuda" and (getattr(self, "param_grad_stream", True) if pg is None else pg == "1"))
        self._pg_env = pg
        force = os.environ.get("VNET_DP_FORCE") == "1" and torch.distributed.is_available() and torch.distributed.is_initialized()
        if self.world > 1 or force:      # VNET_DP_FORCE: run the collective path in a group of one (RCCL on a 1-GPU box)
            parallel.broadcast_parameters(self.flat.data)
            self.optimizer.gscale = 1.0 / self.world
            # launch the bucket all-reduces once this fraction of the gradient bytes exists (parallel.py): keeps the
            # collective off the 256-CU-planned deep-level kernels; TrainingSetting.AllReduceHoldFraction
            # (bf16 mode: the backward pass that is left after encoder level 3 is shorter than the all-reduce -> launch when ready)
            default_hold = 0.99 if ops.get_compute_dtype() in ("fp32", "fp32_split3") else 0.0
            hold = float(getattr(self, "allreduce_hold_fraction", default_hold))
            # two-pass backward (pass 1: output layer, decoder, bottom level = 81 % of the gradient bytes; pass 2: encoder):
            # the replayed step is then gradients graph 1 -> all-reduce of pass 1's buckets (asynchronous) -> gradients graph 2
            # -> the remaining buckets -> optimiser graph, i.e. the collective travels under the encoder's backward kernels
            names = self.flat.names
            enc = ("vnet/encoder", "vnet/input_layer")
            head = [i for i, n in enumerate(names) if not n.startswith(enc)]
            self._two_pass = (os.environ.get("VNET_DP_TWO_PASS", "1") != "0" and hasattr(self.network, "cut_backward") and bool(head)
                              and len(head) < len(names) and head == list(range(len(head))))
            self.sync = parallel.BucketedGradAllReduce(self.flat, hold_fraction=hold, force=force,
                                                       bucket_bytes=int(os.environ.get("VNET_DP_BUCKET_BYTES", 32 << 20)),
                                                       phase1_last=(head[-1] if self._two_pass else None),
                                                       comm_dtype=getattr(self, "grad_comm_dtype", "fp32"))
            if getattr(self, "sync_batch_norm", False):
                # single-device BatchSize = world x per-rank batch semantics of the reference (networks.py:319);
                # the default (per-replica statistics) equals the reference run on each rank's batch alone
                ops.set_sync_batch_norm()
        if self._pg_env is None and not hasattr(self, "param_grad_stream") and (self._graph_mode() != "off" or ops.storage_is_bf16()):
            # the replayed step graph is single-stream (see _build_step_graph); its few eager steps (warm-up, odd batch
            # shapes) then use one stream too, so every launch of the process has the same schedule.  bf16-storage mode is
            # single-stream in every step mode: with the side stream the second gradient of a two-consumer tensor cannot be
            # accumulated by the producing kernel (ops._slot_target) and autograd's add rounds twice (RNE(RNE(a) + RNE(b)) instead
            # of RNE(RNE(a) + b)) -- the eager step would no longer be bit-identical to the replayed one
            ops.set_param_grad_stream(False)

    # -- one training step (reference model.py:743-748: ONE sess.run per step) ------------------------------------
    def _compute_gradients(self, images, labels, dropout, split=False):
        """forward, loss, backward (gradients written into the flat buffer), join the parameter-gradient stream.
        split: stop the backward pass at the network's cut (encoder | bottom level + decoder); _backward_rest() continues."""
        if hasattr(self.network, "cut_backward"):
            self.network.cut_backward = bool(split)
        if getattr(self.network, "fuse_zero_bias_grad", False):
            self.flat.begin_step()            # nothing to clear: see FlatParams.begin_step
        else:
            self.flat.zero_grad()
        ops.begin_dropout_pass()
        _, loss, _, _ = self.forward(images, labels, dropout)
        if getattr(self, "_one", None) is None:
            self._one = torch.ones((), dtype=torch.float32, device=self.device)     # autograd would fill a fresh one every step
        with ops.deferred_wgrad_reduce(self._defer_wgrad_reduce()):
            loss.backward(self._one)
            ops.join_param_grad_stream()      # filter / bias gradients were enqueued on their own stream
        if not torch.cuda.is_current_stream_capturing():
            self.flat.check_accumulation()
        return loss

    def _defer_wgrad_reduce(self):
        """One batched reduce of the filter-gradient slabs at the end of the backward pass instead of one per layer -- unless the
        data-parallel buckets leave from the gradient hooks while backward runs (eager step)."""
        if self.device.type != "cuda":
            return False
        return self.sync is None or bool(self.sync.hold_all)

    def _backward_rest(self):
        """Second backward pass of a split step: from the cut tensors through the encoder."""
        cuts = [(o, l.grad) for o, l in self.network.backward_cuts if l.grad is not None]
        self.network.backward_cuts = []
        with ops.deferred_wgrad_reduce(self._defer_wgrad_reduce()):
            if cuts:
                torch.autograd.backward([o for o, _ in cuts], [g for _, g in cuts])
            ops.join_param_grad_stream()
        if not torch.cuda.is_current_stream_capturing():
            self.flat.check_accumulation()

    def _train_step_eager(self, images, labels, dropout):
        lr = optim.exponential_decay(self.initial_learning_rate, self.global_step, self.decay_steps, self.decay_factor)
        if self.sync is not None:
            self.sync.begin_step()
        if dropout > 0.0 and self.device.type == "cuda":
            # dropout masks from (layer position, step number in the device step state): the same stream of masks
            # whether the step is enqueued kernel by kernel or replayed as a graph
            st = ops.step_state(self.device)
            ops.set_step_state(st, lr, lr, self.global_step)
            with ops.use_step_state(st):
                loss = self._compute_gradients(images, labels, dropout, split=self._two_pass)
                if self._two_pass:
                    self._backward_rest()
        else:
            loss = self._compute_gradients(images, labels, dropout, split=self._two_pass)
            if self._two_pass:
                self._backward_rest()
        if self.sync is not None:
            self.sync.finish()
        self.optimizer.apply(lr)
        self.global_step += 1
        # detached: a caller that keeps the loss must not keep this step's autograd graph (and with it the AccumulateGrad
        # nodes, which remember the stream they were created on -- a later stream capture of the step would then run them
        # on the wrong stream)
        return loss.detach()

    def _graph_mode(self):
        """'off' | 'whole' (single process: the whole step is one hipGraph) | 'segmented' (data parallel: gradients graph ->
        eager RCCL all-reduce of the buckets -> optimiser graph).  No collective is ever captured: torch's
        ProcessGroupNCCL watchdog thread queries the work's events, and an event last recorded in a capturing stream makes
        that query fail with hipErrorCapturedEvent and terminate the process (seen with a captured RCCL all-reduce in a
        group of one).  TrainingSetting.StepGraph / VNET_STEP_GRAPH = 0|1 switches the feature (default on)."""
        if getattr(self, "force_eager", False) or getattr(self, "_graph_failed", False):
            return "off"                       # bench.py's per-launch timing pass (events cannot be timed in a graph) / failed capture
        want = os.environ.get("VNET_STEP_GRAPH")
        want = getattr(self, "step_graph", True) if want is None else want not in ("0", "off", "false")
        if not want or self.device.type != "cuda":
            return "off"
        if self.world > 1 or self.sync is not None:
            if getattr(self, "sync_batch_norm", False):
                return "off"                   # 74 small collectives inside forward/backward: eager only
            return "segmented"
        return "whole"

    def _capture(self, fn, pool=None):
        g = torch.cuda.CUDAGraph()
        # thread_local: the prefetch threads pin host memory (hipHostMalloc) while the main thread captures; in the default
        # "global" mode such a call from ANY thread invalidates the capture (hipErrorStreamCaptureInvalidated)
        with torch.cuda.graph(g, pool=pool, stream=self._graph_stream, capture_error_mode="thread_local"):
            out = fn()
        return g, out

    def _build_step_graph(self, mode, images, labels, dropout):
        """Capture the step on static input buffers.  Everything the host would enqueue (about 400 kernel launches on two
        streams) becomes one hipGraphLaunch; per-step scalars come from the device step state (ops.step_state)."""
        st = self._step_state
        self._g_images = torch.empty_like(images, device=self.device)
        self._g_labels = torch.empty_like(labels, device=self.device)
        self._g_shape = (tuple(images.shape), tuple(labels.shape), float(dropout))
        self._g_images.copy_(images, non_blocking=True)
        self._g_labels.copy_(labels, non_blocking=True)

        def grads():
            with ops.use_step_state(st):
                return self._compute_gradients(self._g_images, self._g_labels, dropout, split=self._two_pass)

        def grads_rest():
            with ops.use_step_state(st):
                self._backward_rest()

        def update():
            self.optimizer.launch(0.0, state=st)

        # Single-stream capture: a graph with the filter-gradient branch on a second stream makes hipGraphLaunch cost as
        # much host time as the eager enqueue (measured 18.8 ms vs 0.11 ms per replay, 128^3 step) and runs 0.9 ms slower
        # on the GPU (the runtime schedules the branch worse than the eager two-stream order), while the side stream
        # itself is worth 0.06 ms -- profiles/r02_step_modes.txt.
        pg_on = self.ctx.pg["on"]
        ops.set_param_grad_stream(False)
        ops.settle_pack_registry()
        try:
            loss = self._capture_mode(mode, grads, update, grads_rest)
        finally:
            ops.set_param_grad_stream(pg_on)
        self._g_loss = loss.detach()
        self._g_mode = mode

    def _capture_mode(self, mode, grads, update, grads_rest=None):
        if mode == "whole":
            def whole():
                loss = grads()
                update()
                return loss
            g, loss = self._capture(whole)
            self._graphs = [g]
        else:
            # (sync.hold_all is set by train_step: hooks only count, reduce_all() launches every bucket after the graph)
            ga, loss = self._capture(grads)
            gr = None
            if self._two_pass:
                gr, _ = self._capture(grads_rest, pool=ga.pool())
            gb, _ = self._capture(update, pool=ga.pool())
            self._graphs = [ga, gb] if gr is None else [ga, gr, gb]
        return loss

    @in_context
    def train_step(self, images, labels, dropout=None):
        """reference model.py:743-748: one fwd + loss + bwd + optimiser step; returns the loss tensor (device scalar; in
        graph mode it is a static buffer that the next step overwrites -- read or clone it before the next call)."""
        dropout = float(self.dropout_rate if dropout is None else dropout)
        mode = self._graph_mode()
        tuner = None
        if mode == "segmented" and getattr(self, "_graphs", None) is not None:       # (after the warm-up steps and the capture)
            tuner = self._dp_tuner()
            if tuner is not None:
                mode = tuner.mode()
                tuner.before()
        if self.sync is not None:
            self.sync.hold_all = (mode in ("segmented", "serial"))       # eager: buckets go out from the hooks, overlapping backward
        if mode == "off":
            loss = self._train_step_eager(images, labels, dropout)
        else:
            loss = self._train_step_graph(mode, images, labels, dropout)
        if tuner is not None:
            tuner.after()
        return loss

    def _all_ranks_agree(self, ok):
        """True iff `ok` on every rank (data parallel: all ranks must enqueue their steps the same way)."""
        if self.world <= 1 or not torch.distributed.is_initialized():
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                            device=self.device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(int(flag.item()))

    @in_context
    def step_mode(self):
        """How train_step currently enqueues the step: 'off' (eager) | 'whole' | 'segmented'."""
        t = getattr(self, "_tuner", None)
        mode = self._graph_mode()
        if mode == "segmented" and t is not None and t.choice is not None:
            return t.choice
        return mode

    def _dp_tuner(self):
        """Data parallel: three interleaved rounds of 5 steps each as segmented graph replay (all-reduce of pass 1's buckets under
        the encoder's backward), as the same graphs with every all-reduce AFTER backward ('serial': the collective never shares
        the CUs with an MFMA kernel whose grid was sized for all 256 -- DESIGN section 5) and as eager steps; 'serial' stays
        unless another mode's median is > 2 % faster (parallel.StepModeAutotune).  TrainingSetting.DpAutotune / VNET_DP_AUTOTUNE = 0 pins the segmented graph."""
        if getattr(self, "_tuner", None) is None:
            on = os.environ.get("VNET_DP_AUTOTUNE")
            on = getattr(self, "dp_autotune", True) if on is None else on not in ("0", "off", "false")
            cands = (["segmented"] + (["serial"] if self._two_pass else []) + ["off"]) if on else ["segmented"]
            pin = os.environ.get("VNET_DP_MODE")               # segmented | serial | off: no measurement, this one
            if pin in cands or pin in ("segmented", "off"):
                cands = [pin]
            # (VNET_DP_AUTOTUNE_STEPS: steps per block, default 5 -- tests whose all-reduce crosses the host over gloo take 1)
            self._tuner = parallel.StepModeAutotune(cands, steps=int(os.environ.get("VNET_DP_AUTOTUNE_STEPS", getattr(self, "dp_autotune_steps", 5))),
                                                    blocks=int(getattr(self, "dp_autotune_blocks", 3)),
                                                    sync=self._device_sync,
                                                    exposed=(self.sync.exposed_seconds if self.sync is not None else None))
        return self._tuner

    def _train_step_graph(self, mode, images, labels, dropout):
        if getattr(self, "_graph_stream", None) is None:
            self._graph_stream = torch.cuda.Stream(device=self.device)
            self._step_state = ops.step_state(self.device)
            self._graphs, self._g_shape, self._g_warm = None, None, 0
        if self._graphs is None and self._g_warm < max(1, int(getattr(self, "step_graph_warmup", 2))):
            # eager steps first: sizes every scratch buffer, builds the packed-filter registry, calibrates the bucket counts
            self._g_warm += 1
            return self._train_step_eager(images, labels, dropout)
        shape = (tuple(images.shape), tuple(labels.shape), dropout)
        serial, mode = (mode == "serial"), ("segmented" if mode == "serial" else mode)       # same graphs, other replay order
        if self._graphs is not None and (shape != self._g_shape or mode != self._g_mode):
            return self._train_step_eager(images, labels, dropout)        # e.g. an odd-sized batch: not the captured shape
        lr = optim.exponential_decay(self.initial_learning_rate, self.global_step, self.decay_steps, self.decay_factor)
        if self._graphs is None:
            torch.cuda.current_stream(self.device).synchronize()
            err = None
            try:
                self._build_step_graph(mode, images, labels, dropout)
            except VnetHipError:
                raise
            except RuntimeError as e:       # a capture the runtime refuses must not end the job: the eager step is the same math
                err = e
            if not self._all_ranks_agree(err is None):
                self._graphs, self._graph_failed = None, True
                self._print("{}: step graph capture failed ({}); continuing with eager steps".format(
                    _now(), str(err).splitlines()[0] if err is not None else "on another rank"))
                return self._train_step_eager(images, labels, dropout)
        else:
            self._g_images.copy_(images, non_blocking=True)
            self._g_labels.copy_(labels, non_blocking=True)
        ops.set_step_state(self._step_state, lr, self.optimizer.schedule(lr), self.global_step)
        self._graphs[0].replay()
        if mode == "segmented":
            if len(self._graphs) == 3 and serial:
                self._graphs[1].replay()
                self.sync.reduce_all()
            elif len(self._graphs) == 3:
                self.sync.reduce_prefix()        # pass 1's buckets (decoder, bottom level) start their all-reduce ...
                self._graphs[1].replay()         # ... while the encoder's backward runs
                self.sync.reduce_rest()          # the rest, then wait for all of them
            else:
                self.sync.reduce_all()           # every bucket: RCCL all-reduce on the communication stream, then wait
            self._graphs[-1].replay()
        self.global_step += 1
        return self._g_loss

    # -- checkpoints (reference model.py:689-702, 758-764, 806-808) ----------------------------------------
    def _ckpt_prefix(self):
        return os.path.join(self.ckpt_dir, "checkpoint")

    @in_context
    def save_checkpoint(self):
        if self.rank != 0:
            return
        os.makedirs(self.ckpt_dir, exist_ok=True)
        path = "%s-%d" % (self._ckpt_prefix(), self.global_step)
        sd = {k: v.cpu() for k, v in self.network.state_dict().items()}
        opt = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in self.optimizer.state_dict().items()}
        torch.save({"variables": sd, "global_step": self.global_step, "start_epoch": self.start_epoch,
                    "optimizer": opt, "opt_names": self.flat.names}, path)
        with open(self._ckpt_prefix() + "-latest", "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))

    @in_context
    def load_checkpoint(self, path=None, with_optimizer=True):
        if path is None:
            with open(self._ckpt_prefix() + "-latest") as f:
                path = os.path.join(self.ckpt_dir, f.readline().split('"')[1])
        if os.path.exists(path + ".index") and not os.path.isfile(path):
            # a checkpoint the REFERENCE wrote (tf.train.Saver: <prefix>.index + .data-*): the same names, another container
            return self.load_tf_checkpoint(path, with_optimizer=with_optimizer)
        ck = torch.load(path, map_location="cpu", weights_only=True)      # tensors, ints and a list of names only
        self.network.load_state_dict(ck["variables"])
        ops.invalidate_packed()
        if self.device is not None and self.device.type == "cuda":
            ops.repack_registered()      # a captured step graph reads the packed filters without checking their tags
        self.global_step, self.start_epoch = int(ck["global_step"]), int(ck["start_epoch"])
        if with_optimizer and getattr(self, "optimizer", None) is not None and ck.get("optimizer"):
            if list(ck.get("opt_names", [])) != list(self.flat.names):
                raise VnetHipError("checkpoint %s: optimiser slots were saved for a different variable layout "
                                   "(%d names vs %d here); restore with the network configuration it was trained with"
                                   % (path, len(ck.get("opt_names", [])), len(self.flat.names)))
            self.optimizer.load_state_dict(ck["optimizer"])
        return path

    @in_context
    def load_tf_checkpoint(self, prefix, with_optimizer=True, verify="small"):
        """Restore from a checkpoint the REFERENCE wrote (`tf.train.Saver`, model.py:689-699, 762, 806: `<ckpt_dir>/checkpoint-<step>` =
        `.index` + `.data-00000-of-00001`): the network's variables by their TF names, global_step, start_epoch and -- when the
        configured optimiser matches -- the Adam (`<var>/Adam`, `<var>/Adam_1`, beta1_power) or Momentum (`<var>/Momentum`) slots.
        tf_checkpoint.py restates the file format; no TensorFlow needed."""
        from . import tf_checkpoint as tfc
        names = list(self.network.state_dict().keys())
        tensors = tfc.read(prefix, verify=verify)
        variables, opt, gs, ep = tfc.split_training_state(tensors, names)
        self.network.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in variables.items()})
        ops.invalidate_packed()
        if self.device is not None and self.device.type == "cuda":
            ops.repack_registered()
        self.global_step, self.start_epoch = gs, ep
        if with_optimizer and opt is not None and getattr(self, "optimizer", None) is not None:
            flat, o = self.flat, self.optimizer

            def fill(dst, table):
                with torch.no_grad():
                    for n, off, p in zip(flat.names, flat.offsets, flat.params):
                        if n in table:
                            dst[off:off + p.numel()].copy_(torch.from_numpy(np.ascontiguousarray(table[n], dtype=np.float32)).reshape(-1).to(dst.device))
            if opt["kind"] == "adam" and hasattr(o, "m"):
                fill(o.m, opt["m"]); fill(o.v, opt["v"]); o.t = opt["t"]
            elif opt["kind"] == "momentum" and hasattr(o, "acc"):
                fill(o.acc, opt["acc"])
        return prefix

    def save_tf_checkpoint(self, prefix=None):
        """Write the training state in the reference's own checkpoint format (what its `saver.restore` reads): variables, optimiser
        slots, global_step, start_epoch.  Pure-Python CRC-32C: a full-width network (176 MB) takes minutes."""
        from . import tf_checkpoint as tfc
        if prefix is None:
            prefix = "%s-%d" % (self._ckpt_prefix(), self.global_step)
        out = {k: v.detach().cpu().numpy() for k, v in self.network.state_dict().items()}
        o = getattr(self, "optimizer", None)
        if o is not None:
            for n, off, p in zip(self.flat.names, self.flat.offsets, self.flat.params):
                sl = slice(off, off + p.numel())
                if hasattr(o, "m"):
                    out[n + "/Adam"] = o.m[sl].view(p.shape).cpu().numpy()
                    out[n + "/Adam_1"] = o.v[sl].view(p.shape).cpu().numpy()
                elif hasattr(o, "acc"):
                    out[n + "/Momentum"] = o.acc[sl].view(p.shape).cpu().numpy()
            if hasattr(o, "m"):
                # TF keeps beta^(t+1) (the power the NEXT step will use) in tf.Variables created under the reference's
                # tf.name_scope("training") (model.py:647-662): `training/beta1_power`, which is what its saver.restore looks up
                out["training/beta1_power"] = np.float32(o.b1 ** (o.t + 1))
                out["training/beta2_power"] = np.float32(o.b2 ** (o.t + 1))
        out["global_step"] = np.int64(self.global_step)
        out["start_epoch"] = np.array([self.start_epoch], dtype=np.int32)
        tfc.write(prefix, out)
        # the state file tf.train.latest_checkpoint(ckpt_dir, latest_filename="checkpoint-latest") reads (model.py:696-699)
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint-latest"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
        return prefix

    def _dataset(self, data_dir, train):
        tf = None
        if self.training_pipeline:
            # TrainingSetting.Pipeline: the reference's transform YAML (model.py:340-372); the index / intensity transforms
            # are restated on arrays (vnet_tensorflow_amd/transforms.py), SimpleITK resampling ones are refused
            from . import transforms as vtf
            tf = vtf.build_pipeline(self.training_pipeline, "train" if train else "test")
        return vdata.VolumeDataset(data_dir, self.image_filenames, self.label_filename, self.label_classes,
                                   self.patch_shape, self.batch_size, train=train, synthetic=self.synthetic,
                                   rank=self.rank, world=self.world, transforms=tf)

    # -- reference model.py:632-815 ---------------------------------------------------------------------------
    @in_context
    def train(self):
        self._print("{}: VNet training start...".format(_now()))
        self.rank, self.local_rank, self.world = parallel.init_from_env()
        self.read_config()
        self.build_model_graph()
        self._setup_training()
        if not self.restore_training:
            if self.rank == 0:
                for d in (self.log_dir, self.ckpt_dir):
                    if os.path.exists(d):
                        shutil.rmtree(d)
                    os.makedirs(d)
        elif os.path.exists(self._ckpt_prefix() + "-latest"):
            self._print("{}: Last checkpoint found at {}, loading...".format(_now(), self.ckpt_dir))
            self.load_checkpoint()
            self._print("{}: Last checkpoint epoch: {}".format(_now(), self.start_epoch))
            self._print("{}: Last checkpoint global step: {}".format(_now(), self.global_step))
        train_set = self._dataset(self.train_data_dir, True)
        test_iter = None
        if self.testing:
            test_set = self._dataset(self.test_data_dir, False)
            test_iter = iter(test_set)
        prefetch = int(getattr(self, "prefetch_depth", 4))
        workers = int(getattr(self, "loader_threads", 3))
        self.steps_timed, self.seconds_timed = 0, 0.0

        first_epoch = self.start_epoch
        for epoch in range(self.start_epoch, self.epoches):
            self._print("{}: Epoch {} starts...".format(_now(), epoch + 1))
            loss_sum, count = 0.0, 0
            pending = None                    # (pinned host scalar, event): the loss of the step before, still in flight
            loader = vdata.Prefetcher(train_set, depth=prefetch, workers=workers) if prefetch > 0 else train_set
            feeder = _DeviceFeeder(self.device, loader)

            def flush(p):
                # the reference prints every step's loss (model.py:749); reading it one step late keeps the host one
                # step ahead of the GPU instead of draining the queue on every iteration
                if p is None:
                    return 0.0, 0
                host, ev = p
                if ev is not None:
                    ev.synchronize()
                val = float(host)
                self.last_loss = val
                self._print('{}: Segmentation training loss: {}'.format(_now(), str(val)))
                return val, 1

            t_epoch = None
            for image, label in feeder:
                if self.global_step > self.max_itr:
                    flush(pending)
                    self._print("{}: Reach maximum iteration steps, training abort.".format(_now()))
                    return
                loss_t = self.train_step(image, label)
                feeder.step_done()
                if self.device.type == "cuda":
                    host = torch.empty((), dtype=torch.float32, pin_memory=True)
                    host.copy_(loss_t.detach(), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    now = (host, ev)
                else:
                    now = (loss_t.detach().clone(), None)
                v, n = flush(pending)
                loss_sum += v
                count += n
                pending = now
                # steady-state throughput: past the capture / warm-up steps and, when there are several epochs, past the
                # first one (which also fills the dataset's volume cache)
                if t_epoch is None and self.global_step >= 5 and (epoch > first_epoch or self.epoches - first_epoch == 1):
                    self._device_sync()
                    t_epoch, s_epoch = time.perf_counter(), self.global_step
                if self.global_step % self.log_interval == 0:
                    self._print("{}: Saving checkpoint of step {} at {}...".format(_now(), self.global_step, self.ckpt_dir))
                    self.save_checkpoint()
                if self.testing and (self.global_step % self.test_step == 0):
                    try:
                        timage, tlabel = next(test_iter)
                    except StopIteration:
                        test_iter = iter(test_set)
                        timage, tlabel = next(test_iter)
                    with torch.no_grad():
                        tl = torch.from_numpy(tlabel).to(self.device)
                        _, tloss, tsm, tpred = self.forward(torch.from_numpy(timage).to(self.device), tl, 0.0, want_softmax=True, want_pred=True)
                        # accuracy + streaming per-class tp/tn/fp/fn, sensitivity, specificity, hard dice, ROC AUC
                        # (tf.metrics.*, reference model.py:588-626)
                        if getattr(self, "metrics", None) is None:
                            self.metrics = ops.StreamingMetrics(self.output_channel_num)
                        self.last_metrics = self.metrics.update(tpred, tl[..., 0], tsm).result()
                    self._print('{}: Segmentation testing accuracy: {:.4f} dice: {}'.format(
                        _now(), self.last_metrics["accuracy"],
                        [round(self.last_metrics[c]["dice"], 4) for c in range(self.output_channel_num)]))
                    self._print("{}: Segmentation testing loss: {}".format(_now(), str(float(tloss.detach()))))
            v, n = flush(pending)
            loss_sum += v
            count += n
            if t_epoch is not None:
                self._device_sync()
                self.steps_timed += self.global_step - s_epoch
                self.seconds_timed += time.perf_counter() - t_epoch
            self._print("{}: Training of epoch {} complete, epoch loss: {}".format(_now(), epoch + 1, loss_sum / max(count, 1)))
            self.start_epoch += 1
            self._print("{}: Saving checkpoint of epoch {} at {}...".format(_now(), epoch + 1, self.ckpt_dir))
            self.save_checkpoint()

    @in_context
    def _infer(self, batch):
        """softmax of one evaluation batch (model.py:914-917: dropout 0, batch statistics of THIS batch).  Enqueued kernel by
        kernel: a replayed hipGraph of the forward pass was built and measured 5-8 % SLOWER end to end here
        (profiles/bench_infer.py: 66.7 vs 70.3 patches/s fp32, 109 vs 118 bf16) -- the eager enqueue already runs ahead of
        the GPU while the next batch is cropped and copied on other threads / streams."""
        with torch.no_grad():
            return self.forward(batch, None, 0.0)[2]

    # -- reference model.py:817-977 (array in / arrays out; SimpleITK resampling not carried) -----------------
    @in_context
    def evaluate_single_3D(self, images_np):
        """images_np float32 [X,Y,Z,Cin] -> (label int64 [X,Y,Z], softmax float32 [K,X,Y,Z]).
        Patch enumeration, the duplicated last batch and argmax-of-summed-softmax follow
        model.py:866-937 exactly; accumulation runs on the GPU."""
        ps, st = list(self.patch_shape), list(self.evaluate_stride)
        pads = [(0, max(p - s, 0)) for s, p in zip(images_np.shape[:3], ps)]
        orig = images_np.shape[:3]
        if any(hi for _, hi in pads):
            images_np = np.pad(images_np, pads + [(0, 0)])
        dims = images_np.shape[:3]
        nums = [int(math.ceil((dims[a] - ps[a]) / float(st[a]))) + 1 for a in range(3)]
        batches, tmp, patch_total = [], [], 0
        for i in range(nums[0]):
            for j in range(nums[1]):
                for k in range(nums[2]):
                    if patch_total % self.evaluate_batch == 0:
                        tmp = []
                    idx = []
                    for a, n in zip(range(3), (i, j, k)):
                        s0 = n * st[a]
                        if s0 + ps[a] > dims[a]:
                            s0 = dims[a] - ps[a]
                        idx += [s0, s0 + ps[a]]
                    tmp.append(idx)
                    if patch_total % self.evaluate_batch == 0:
                        batches.append({'images': images_np, 'indexes': tmp})
                    patch_total += 1
        batches.append({'images': images_np, 'indexes': tmp})        # "for last batch" (model.py:903)
        K = self.output_channel_num
        vol = torch.zeros(dims + (K,), dtype=torch.float32, device=self.device)
        cnt = torch.zeros(dims, dtype=torch.float32, device=self.device)
        # patches are cropped on worker threads into pinned memory and copied to the device on the copy stream while the
        # previous batch is in the network (the same feeder as the training loop)
        def crop(bd):
            t = torch.from_numpy(prepare_batch(bd))
            return (t.pin_memory() if self.device.type == "cuda" else t), torch.zeros(1, dtype=torch.int32)

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(getattr(self, "loader_threads", 3))) as pool:
            def windowed(depth=3):
                # a bounded window of crop jobs in flight (Executor.map would submit -- and pin -- every patch of the volume at once)
                from collections import deque
                q, it = deque(), iter(batches)
                for bd in it:
                    q.append(pool.submit(crop, bd))
                    if len(q) >= depth:
                        yield q.popleft().result()
                while q:
                    yield q.popleft().result()
            feeder = _DeviceFeeder(self.device, windowed())
            for bd, (batch, _) in zip(batches, feeder):
                sm = self._infer(batch)
                for j, idx in enumerate(bd['indexes']):
                    ops.accumulate_patch(sm[j], vol, cnt, (idx[0], idx[2], idx[4]))
                feeder.step_done()
        # argmax of the summed softmax (model.py:934) on the device: only the label map crosses PCIe unless probabilities are wanted
        label_np = torch.argmax(vol, dim=-1).to(torch.int16 if K < 32768 else torch.int64).cpu().numpy().astype(np.int64)
        sl = tuple(slice(0, s) for s in orig)
        if not self.evaluate_probability_output:
            return label_np[sl], None
        vol_np, cnt_np = vol.cpu().numpy(), cnt.cpu().numpy()
        softmax_np = np.moveaxis(vol_np, -1, 0) / np.float32(cnt_np)          # model.py:935-937
        return label_np[sl], softmax_np[(slice(None),) + sl]

    # -- reference model.py:1131-1242 ------------------------------------------------------------------------------
    @in_context
    def evaluate(self):
        self.read_config()
        self.rank, self.local_rank, self.world = 0, 0, 1
        self.build_model_graph()
        self._print("{}: Restoring checkpoint {}".format(_now(), self.checkpoint_path))
        self.load_checkpoint(self.checkpoint_path, with_optimizer=False)
        for case in sorted(os.listdir(self.evaluate_data_dir)):
            cdir = os.path.join(self.evaluate_data_dir, case)
            if not os.path.isdir(cdir):
                continue
            chans = [np.asarray(vdata.load_volume(os.path.join(cdir, f)), dtype=np.float32) for f in self.evaluate_image_filenames]
            image = np.stack(chans, axis=-1)
            if self.evaluate_pipeline:
                # EvaluationSetting.Pipeline: the 'evaluate' transform list of the reference's YAML (model.py:1142-1167);
                # intensity transforms and Padding are restated on arrays, physical-grid resampling is refused
                from . import transforms as vtf
                tf = vtf.build_pipeline(self.evaluate_pipeline, "evaluate")
                image, _ = vtf.apply_pipeline(tf, image, np.zeros(image.shape[:3], dtype=np.int32), np.random.default_rng(0))
            label, softmax = self.evaluate_single_3D(image)
            label = label[tuple(slice(0, n) for n in chans[0].shape)]
            if softmax is not None:
                softmax = softmax[(slice(None),) + tuple(slice(0, n) for n in chans[0].shape)]
            # physical voxel size of the input (the reference compares GetPhysicalSize against VolumeThreshold, model.py:117-140)
            spacing = vdata.volume_spacing(os.path.join(cdir, self.evaluate_image_filenames[0]))
            if self.evaluate_lcc:                                     # model.py:1218-1219
                label = ExtractLargestConnectedComponents(label, spacing)
            if self.evaluate_volume_threshold and self.evaluate_volume_threshold > 0:      # model.py:1222-1223
                label = volume_threshold(label, self.evaluate_volume_threshold, spacing)
            out = os.path.join(cdir, self.evaluate_label_filename)
            if out.endswith(".npy"):
                np.save(out, label.astype(np.int16))
            else:
                vdata.write_nifti(out[:-3] if out.endswith(".gz") else out, label.astype(np.int16), spacing)
            if self.evaluate_probability_output:
                for c in range(softmax.shape[0]):
                    name = self.evaluate_probability_filename
                    stem, ext = (name[:-7], ".nii") if name.endswith(".nii.gz") else os.path.splitext(name)
                    pout = os.path.join(cdir, "%s_%s%s" % (stem, str(self.label_classes[c]), ext))
                    if ext == ".npy":
                        np.save(pout, softmax[c])
                    else:
                        vdata.write_nifti(pout, softmax[c].astype(np.float32), spacing)
            self._print("{}: Evaluation of {} complete".format(_now(), case))

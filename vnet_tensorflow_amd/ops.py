"""torch.autograd bridges over the C ABI of libvnet_hip.so.

PyTorch is plumbing only here: it owns device memory, streams and the autograd tape; every
forward and backward computation below is a call into the HIP library on the current stream.
There is no CPU or eager-PyTorch fallback -- non-GPU tensors raise (meta tensors are accepted
for shape inference only, so a network can create its variables before the first batch).
"""
import contextlib
import ctypes
import math
import os as _os
import weakref

import torch

from . import _lib
from ._lib import VnetHipError, check

BN_EPS = 1e-3        # reference networks.py:259 epsilon=0.001
BN_MOMENTUM = 0.99   # reference networks.py:259 momentum=0.99

ACT = {None: 0, "none": 0, "relu": 1, "prelu": 2, "lrelu": 3}
PACK_FWD, PACK_BWD, PACK_UP, PACK_FWD_BF16, PACK_BWD_BF16, PACK_BOTH_BF16, PACK_BOTH = 0, 1, 2, 3, 4, 5, 6
PACK_FWD_X3, PACK_BWD_X3, PACK_BOTH_X3 = 7, 8, 9

# Arithmetic of the 5x5x5 convolutions (forward, backward-data and filter gradient): "fp32" = exact fp32 MFMA (the reference's
# arithmetic), "bf16" = operands rounded to bf16, fp32 accumulation (BASELINE config C5).  Everything else
# (2^3 down/up convolutions, the fused 1-channel input block, batch-norm, loss, optimiser) is fp32 in both modes.
#   "bf16"          (round 3, BASELINE config C5 as SURVEY 8(d) states it): bf16 STORAGE -- activations, skip tensors and their
#                   gradients are torch.bfloat16 tensors, every kernel on them computes in fp32 and rounds once (include/vnet_hip.h,
#                   `*_b16`); 5^3 AND 2^3 convolutions take bf16 operands; statistics, parameter gradients, logits and loss fp32.
#                   The ops below dispatch on the tensor dtype, so the mode only decides what the network input is turned into.
#   (round 2's "bf16_operands" -- fp32 tensors + bf16 shadows, bf16 operands in the 5^3 convolutions only -- was superseded by the
#   storage mode in round 3 and retired in round 5 together with its `*_x16` / `vnet_conv_*_bf16` entry points.)
# ---- per-model state (round 4) ---------------------------------------------------------------------------------------------
# What used to be process globals -- the compute dtype, the parameter-gradient stream and the registry of packed filters -- lives
# in an OpsContext.  image2label owns one and enters it around everything it runs (model.in_context), so a fp32 and a bf16 model
# can coexist in one process and interleave their steps; code that calls the ops directly (tests, tools) works on the default
# context exactly as before.  The switch is per process, not per thread: models take turns, they do not run concurrently.
class OpsContext(object):
    def __init__(self):
        self.compute = {"dtype": "fp32", "store16": False, "name": "fp32", "split3": False}
        self.pg = {"on": False, "streams": {}, "used": set(), "keep": []}
        self.pack_epoch = [0]
        self.pack_reg = {"entries": [], "descs": None, "device": None}     # every (filter, mode) ever packed: repacked in one launch


_DEFAULT_CTX = OpsContext()
_CTX = [_DEFAULT_CTX]
_TEST = {"test_delay": 0}          # (tests) cycles the parameter-gradient stream is held back per layer; shared by every context


class context(object):
    """with ops.context(ctx): ... -- the ops inside use ctx's compute dtype, parameter-gradient stream and pack registry."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.prev = _CTX[0]
        _CTX[0] = self.ctx
        return self.ctx

    def __exit__(self, *a):
        _CTX[0] = self.prev


def current_context():
    return _CTX[0]


class _CtxDict(object):
    """dict-like view of one attribute of the CURRENT context (keeps the module-level names the ops were written with)."""

    def __init__(self, attr, shared=()):
        self._attr, self._shared = attr, shared

    def _d(self, k=None):
        return _TEST if k in self._shared else getattr(_CTX[0], self._attr)

    def __getitem__(self, k):
        return self._d(k)[k]

    def __setitem__(self, k, v):
        self._d(k)[k] = v

    def __delitem__(self, k):
        del self._d(k)[k]

    def __contains__(self, k):
        return k in self._d(k)

    def get(self, k, default=None):
        return self._d(k).get(k, default)


_COMPUTE = _CtxDict("compute")
PACK_ROUND16 = 16


def set_compute_dtype(dtype):
    if dtype not in ("fp32", "fp32_split3", "bf16"):
        raise VnetHipError("compute dtype must be 'fp32', 'fp32_split3' or 'bf16', got %r" % (dtype,))
    _COMPUTE["dtype"] = "fp32" if dtype in ("fp32", "fp32_split3") else "bf16"
    _COMPUTE["store16"] = dtype == "bf16"
    # "fp32_split3" (round 5, csrc/conv_x3.h): fp32 tensors and fp32 accuracy, but the 5^3 convolutions form every product from six
    # bf16 products of exactly split operands (x = h + m + l) on the bf16 matrix pipe; layers the f32x3 kernels do not take
    # (vnet_conv_x3_ok) keep the fp32 MFMA kernels
    _COMPUTE["split3"] = dtype == "fp32_split3"
    _COMPUTE["name"] = dtype


def get_compute_dtype():
    return _COMPUTE["name"]


def storage_is_bf16():
    return _COMPUTE["store16"]


def _is16(t):
    return t is not None and t.dtype == torch.bfloat16
LOSS_KIND = {"sorensen": 0, "jaccard": 1, "xent": 2}
LOSS_WEIGHTED, LOSS_MIXED = 16, 32


def _ptr(t):
    return t.data_ptr() if t is not None else None      # ctypes converts an int to the declared c_void_p itself


_LAUNCH_ON = [None]      # raw handle of the stream the next launches go to instead of torch's current one (parameter-gradient stream)


def _raw_stream():
    """Handle of the stream kernels are launched on: torch's current stream on the current device, unless the
    parameter-gradient section redirected launches.  (torch.cuda.current_stream() builds a Stream object through three Python
    layers, ~8 us, and a torch.cuda.stream() context costs ~20 us; this is called for every kernel launch.)"""
    h = _LAUNCH_ON[0]
    return h if h is not None else torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return _raw_stream()


# ---- device-resident step state (include/vnet_hip.h: vnet_step_state_set) ---------------------------------
# A captured hipGraph freezes kernel arguments; the learning rate, Adam's bias-corrected lr_t and the dropout stream
# position live in a 32-byte device buffer that one tiny eager kernel rewrites before each replay.
_STEP_STATE = {"buf": {}, "active": None}


def step_state(device):
    buf = _STEP_STATE["buf"].get(device)
    if buf is None:
        buf = torch.zeros(32, dtype=torch.uint8, device=device)
        _STEP_STATE["buf"][device] = buf
    return buf


def set_step_state(state, lr, lr_t, step):
    check(_lib.lib().vnet_step_state_set(_ptr(state), float(lr), float(lr_t), int(step), _stream()), "vnet_step_state_set")


@contextlib.contextmanager
def use_step_state(state):
    """Inside: dropout takes its per-step seed offset from `state` (graph-replayable) instead of a host counter."""
    prev = _STEP_STATE["active"]
    _STEP_STATE["active"] = state
    try:
        yield
    finally:
        _STEP_STATE["active"] = prev


def _need_gpu(t, what, allow16=False):
    if not t.is_cuda:
        raise VnetHipError("%s: tensor on %s -- the HIP library is the only compute path (no CPU fallback)" % (what, t.device))
    if t.dtype != torch.float32 and not (allow16 and t.dtype == torch.bfloat16):
        raise VnetHipError("%s: expected float32%s, got %s" % (what, " or bfloat16" if allow16 else "", t.dtype))


# ---- workspace (caller-owned, per device) -----------------------------------------------------
_WS = {}
_WS_RETIRED = []


def workspace(nbytes, device):
    """One scratch buffer per (device, stream): launches on the parameter-gradient stream never share scratch
    with the main stream's launches."""
    nbytes = int(nbytes)
    key = (device, _raw_stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # never hand a scratch buffer back: launches on a redirected stream may still be using it (the allocator files the
            # block under torch's current stream), and a captured step graph replays launches that hold its ADDRESS (ADVICE r2:
            # an eager step that needs more scratch after the capture would otherwise free memory the graph still writes)
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(nbytes, 1 << 20) * 5 // 4, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ---- parameter-gradient stream --------------------------------------------------------------------------
# The backward critical path is  batch-norm backward -> backward-data conv -> next layer; filter and bias gradients
# only feed the optimiser / the gradient all-reduce.  With this switch on they are enqueued on a second HIP stream,
# so their MFMA work fills the chip while the critical path sits in its short HBM- and latency-bound kernels.
# Only gradients that go straight into the flat gradient buffer (GradSink) take this route.
_PG = _CtxDict("pg", shared=("test_delay",))


def set_param_grad_stream(on):
    _PG["on"] = bool(on)


def param_grad_stream(device, create=True):
    if not _PG["on"]:
        return None
    st = _PG["streams"].get(device)
    if st is None and create:
        st = torch.cuda.Stream(device=device)
        _PG["streams"][device] = st
    return st


def join_param_grad_stream(device=None):
    """Make the current stream wait for every parameter gradient enqueued so far (call before reading gradients)."""
    for dev in list(_PG["used"]):
        if device is None or dev == device:
            torch.cuda.current_stream(dev).wait_stream(_PG["streams"][dev])
            _PG["used"].discard(dev)
    if not _PG["used"]:
        _PG["keep"] = []


# ---- packed-weight cache -------------------------------------------------------------------------
_PACK_EPOCH = _CtxDict("pack_epoch")
_PACK_REG = _CtxDict("pack_reg")


def invalidate_packed():
    """Call after parameters change outside autograd's version counter (optimiser kernels)."""
    _PACK_EPOCH[0] += 1


def _pack_tag(w):
    return (_PACK_EPOCH[0], w._version, w.data_ptr())


def packed_weights(w, mode, taps, I, O):
    """MFMA-fragment-ordered copy of a filter, cached ON the tensor object (a cache keyed by
    data_ptr would go stale when the allocator recycles an address)."""
    cache = getattr(w, "_vnet_packed", None)
    if cache is None:
        cache = {}
        w._vnet_packed = cache
    key = (mode, taps, I, O)
    ent = cache.get(key)
    tag = _pack_tag(w)
    if ent is not None and ent[0] == tag:
        return ent[1]
    L = _lib.lib()
    if ent is None:
        wp = torch.empty(L.vnet_packed_weight_floats(mode, taps, I, O), dtype=torch.float32, device=w.device)
        if isinstance(w, torch.nn.Parameter):                  # long-lived filter: join the batched repack
            _PACK_REG["entries"].append((weakref.ref(w), key, wp))
            _PACK_REG["descs"] = None
    else:
        wp = ent[1]
    check(L.vnet_pack_weights(mode, _ptr(w), _ptr(wp), taps, I, O, _stream()), "vnet_pack_weights")
    cache[key] = (tag, wp)
    return wp


_PACK_BOTH = {"on": True}      # (tests / A-B scripts flip the entry)


def repack_registered():
    """After an optimiser step: refresh the packed copy of every registered filter in ONE kernel launch
    (instead of ~58 small launches spread over the next forward/backward pass)."""
    reg = _PACK_REG
    alive = [(r(), key, wp) for r, key, wp in reg["entries"]]
    if any(w is None for w, _, _ in alive):                     # networks that were garbage-collected
        reg["entries"] = [(r, key, wp) for (r, key, wp), (w, _, _) in zip(reg["entries"], alive) if w is not None]
        reg["descs"] = None
    ents = [e for e in alive if e[0] is not None]
    if not ents:
        return
    L = _lib.lib()
    dev = ents[0][0].device
    ptrs = tuple(w.data_ptr() for w, _, _ in ents)
    if reg["descs"] is None or reg.get("ptrs") != ptrs:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise VnetHipError("the packed-filter registry changed inside a stream capture (a network was garbage-collected or "
                               "a new filter appeared): call ops.settle_pack_registry() before capturing")
        rows = []
        # a filter with BOTH bf16 images registered (forward + backward-data: every 5^3 filter of a bf16 training step) and whole
        # 32-channel blocks: one descriptor, one read of w for the two images (VNET_PACK_BOTH_BF16; -0.06 ms per C5 step)
        both = {}
        fams = {PACK_FWD_BF16: 0, PACK_BWD_BF16: 0, PACK_FWD: 1, PACK_BWD: 1,       # (the plain fp32 pair the same way: VNET_PACK_BOTH;
                PACK_FWD_X3: 2, PACK_BWD_X3: 2}                                    #  the f32x3 pair: VNET_PACK_BOTH_X3, round 6)
        if _os.environ.get("VNET_PACK_BOTH_X3", "1") == "0":
            fams.pop(PACK_FWD_X3); fams.pop(PACK_BWD_X3)
        if _PACK_BOTH["on"]:
            for w, (mode, taps, I, O), wp in ents:
                if mode in fams and I % 32 == 0 and O % 32 == 0:
                    both.setdefault((w.data_ptr(), fams[mode]), {})[mode] = wp
        done = set()
        for w, (mode, taps, I, O), wp in ents:
            key = (w.data_ptr(), fams.get(mode))
            pair = both.get(key, {})
            if mode in fams and len(pair) == 2 and I % 32 == 0 and O % 32 == 0:
                if key not in done:
                    done.add(key)
                    f, bk, mb = ((PACK_FWD_BF16, PACK_BWD_BF16, PACK_BOTH_BF16), (PACK_FWD, PACK_BWD, PACK_BOTH),
                                 (PACK_FWD_X3, PACK_BWD_X3, PACK_BOTH_X3))[fams[mode]]
                    rows.append([w.data_ptr(), pair[f].data_ptr(), mb, taps, I, O, pair[bk].data_ptr(), 0])
                continue
            cq, npad = ctypes.c_int(), ctypes.c_int()
            check(L.vnet_packed_dims(mode, taps, I, O, ctypes.byref(cq), ctypes.byref(npad)), "vnet_packed_dims")
            rows.append([w.data_ptr(), wp.data_ptr(), mode, taps, I, O, cq.value, npad.value])
        reg["descs"] = torch.tensor(rows, dtype=torch.int64).to(dev)
        reg["nrows"] = len(rows)
        reg["ptrs"] = ptrs
    check(L.vnet_pack_weights_batched(_ptr(reg["descs"]), reg["nrows"], _stream()), "vnet_pack_weights_batched")
    for w, key, wp in ents:
        w._vnet_packed[key] = (_pack_tag(w), wp)


def settle_pack_registry():
    """Before a stream capture: collect garbage (torch.cuda.graph does it on entry anyway -- a network that dies THERE would
    change the registry inside the capture), drop dead filters and rebuild the descriptor table now, while host-to-device
    copies are still allowed."""
    import gc
    gc.collect()
    repack_registered()


def clear_pack_registry():
    _PACK_REG["entries"] = []
    _PACK_REG["descs"] = None


def _same_out(n, s):
    return -(-n // s)


# ---- gradient sinks ---------------------------------------------------------------------------------
class GradSink(object):
    """Registered on a Parameter by optim.FlatParams: backward kernels write that parameter's gradient
    straight into its slice of the flat gradient buffer (no temporary, no autograd accumulate kernel) and
    then call `ready` (the data-parallel bucket countdown).  `written` guards against a parameter that is
    used twice in one backward pass: the second use falls back to autograd's accumulation."""
    __slots__ = ("view", "written", "ready", "ws")

    def __init__(self, view):
        self.view, self.written, self.ready = view, False, None
        self.ws = None           # this parameter's own slab buffer of the deferred filter-gradient reduce (lives as long as the sink)


def _grad_out(p):
    s = getattr(p, "_vnet_sink", None)
    if s is not None and not s.written:
        return s.view, s
    return torch.empty_like(p), None


def _grad_ret(t, s):
    if s is None:
        return t
    s.written = True
    if s.ready is not None:
        s.ready()
    return None


# ---- optional per-launch timing (bench.py): HIP events on the launch stream ---------------------
_PROFILE = {"on": False, "records": [], "only": None}


def profile_start(only=None):
    """`only`: set of launch tags to time (None = every conv-family launch).  Each timed launch costs two event
    packets on the stream (~5.6 us of idle GPU each on MI355X), so a throughput run times only the kernels it reports."""
    _PROFILE["records"] = []
    _PROFILE["only"] = set(only) if only is not None else None
    _PROFILE["on"] = True


def profile_stop():
    """Returns [(tag, flops, algorithmic_bytes, milliseconds)] after synchronising."""
    _PROFILE["on"] = False
    torch.cuda.synchronize()
    out = [(t, f, b, e0.elapsed_time(e1)) for (t, f, b, e0, e1) in _PROFILE["records"]]
    _PROFILE["records"] = []
    return out


def _timed_tag(tag):
    return _PROFILE["on"] and (_PROFILE["only"] is None or tag in _PROFILE["only"])


def _wgrad_tag(bf16, ks, kx, stride, wo, B, cin, co):
    return "wgrad%s k%d%s s%d %d^3x%d %d->%d" % ("-bf16" if bf16 else "", ks, "x%d" % kx if kx else "", stride, wo, B, cin, co)


class _Timed(object):
    """HIP events around one launch.  Not under stream capture: on this runtime (ROCm 7.0 libamdhip64 bundled with torch
    2.10) an event recorded into a capture cannot be timed afterwards -- hipEventRecordWithFlags(hipEventRecordExternal)
    is refused with hipErrorInvalidValue and hipEventElapsedTime on captured events returns hipErrorInvalidHandle
    (profiles/probes/graph_event_probe.py) -- so bench.py times the reported kernel family in eager steps."""

    def __init__(self, tag, flops, nbytes):
        self.rec = (tag, flops, nbytes)
        self.on = _timed_tag(tag) and not torch.cuda.is_current_stream_capturing()

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()          # torch's current stream == the stream the kernel is launched on
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            _PROFILE["records"].append(self.rec + (self.e0, self.e1))


# ---- convolution family ----------------------------------------------------------------------------
def _conv_call(ks, stride, up, x0, x1, wp, bias, y0, y1, dims_in, dims_out, kx=0, accum=False, stats=None, res=None):
    L = _lib.lib()
    B = x0.shape[0]
    C0, C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    Cy0, Cy1 = y0.shape[-1], (y1.shape[-1] if y1 is not None else 0)
    nb = L.vnet_conv_ws_bytes(ks, kx, stride, up, C0 + C1, Cy0 + Cy1, B, *dims_out)
    ws = workspace(nb, x0.device) if nb else None
    nin, nout = B * dims_in[0] * dims_in[1] * dims_in[2], B * dims_out[0] * dims_out[1] * dims_out[2]
    taps = 8 if up else ks * ks * (kx or ks)
    mac_vox = nin if up else nout          # the transposed conv does its 8 taps per INPUT voxel
    flops = 2.0 * mac_vox * taps * (C0 + C1) * (Cy0 + Cy1)
    nbytes = 4.0 * (nin * (C0 + C1) + nout * (Cy0 + Cy1) + taps * (C0 + C1) * (Cy0 + Cy1) + (Cy0 + Cy1))
    tag = "conv k%d%s s%d%s %d^3x%d %d->%d" % (ks, "x%d" % kx if kx else "", stride, " up" if up else "", dims_out[2], B, C0 + C1, Cy0 + Cy1)
    with _Timed(tag, flops, nbytes):
        if stats is not None:        # batch-norm statistics of y (+ res) from the epilogue
            check(L.vnet_conv_fwd_stats(ks, kx, stride, _ptr(x0), C0, _ptr(x1), C1, _ptr(wp), _ptr(bias), _ptr(y0), Cy0,
                                        B, *dims_in, *dims_out, _ptr(res), _ptr(stats), _ptr(ws), nb, _stream()), "vnet_conv_fwd_stats")
            return
        fn = L.vnet_conv_fwd_acc if accum else L.vnet_conv_fwd
        check(fn(ks, kx, stride, up, _ptr(x0), C0, _ptr(x1), C1, _ptr(wp), _ptr(bias),
                 _ptr(y0), Cy0, _ptr(y1), Cy1, B, *dims_in, *dims_out,
                 _ptr(ws), nb, _stream()), "vnet_conv_fwd")


# force: (tests) take the f32x3 kernels for every shape they can run, not only where they pay;  ksplit: the deep levels (few bricks:
# channel chunks split over workgroups, partial slabs + reduce) take them too (`_X3["ksplit"] = False`: A/B measurements)
_X3 = {"force": False, "ksplit": True}


def _x3_ok(C0, C1, Cy0, Cy1, B, dims):
    if not _COMPUTE.get("split3"):
        return False
    if _X3["force"]:
        return C0 % 16 == 0 and C1 % 16 == 0 and Cy0 % 16 == 0 and Cy1 % 16 == 0
    L = _lib.lib()
    if L.vnet_conv_x3_ok(C0, C1, Cy0, Cy1, B, *dims) != 1:
        return False
    return _X3["ksplit"] or L.vnet_conv_x3_ws_bytes(C0 + C1, Cy0 + Cy1, B, *dims) == 0


def _conv_x3_call(x0, x1, wp, bias, y0, y1, dims, accum=False, stats=None, res=None):
    """5^3 stride-1 conv, fp32 in / out, products from three-way split bf16 operands (vnet_conv_fwd_x3)."""
    L = _lib.lib()
    B = x0.shape[0]
    C0, C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    Cy0, Cy1 = y0.shape[-1], (y1.shape[-1] if y1 is not None else 0)
    nvox = B * dims[0] * dims[1] * dims[2]
    flops = 2.0 * nvox * 125 * (C0 + C1) * (Cy0 + Cy1)
    nbytes = 4.0 * (nvox * (C0 + C1) + nvox * (Cy0 + Cy1) + 125 * (C0 + C1) * (Cy0 + Cy1) + (Cy0 + Cy1))
    tag = "conv-x3 k5 s1 %d^3x%d %d->%d" % (dims[2], B, C0 + C1, Cy0 + Cy1)
    nb = L.vnet_conv_x3_ws_bytes(C0 + C1, Cy0 + Cy1, B, *dims)
    ws = workspace(nb, x0.device) if nb else None
    with _Timed(tag, flops, nbytes):
        check(L.vnet_conv_fwd_x3(_ptr(x0), C0, _ptr(x1), C1, _ptr(wp), _ptr(bias), _ptr(y0), Cy0, _ptr(y1), Cy1, B, *dims,
                                 _ptr(y0) if accum else None, _ptr(res), _ptr(stats), _ptr(ws), nb, _stream()), "vnet_conv_fwd_x3")


# ---- bf16-storage convolution calls (include/vnet_hip.h: vnet_conv_fwd_b16, vnet_conv2_fwd_b16, ...) ----------------------------
_IN4 = {"on": True}


def _conv5_b16_call(x0, x1, wp, bias, y0, y1, dims, accum=False, stats=None, res=None, acc_src=None, cin_real=0):
    """cin_real: the input channels of the FILTER when x0 carries zero-padded channels behind them (the cast network input)."""
    L = _lib.lib()
    B = x0.shape[0]
    C0, C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    Cy0, Cy1 = y0.shape[-1], (y1.shape[-1] if y1 is not None else 0)
    nb = L.vnet_conv_b16_ws_bytes(C0, C1, Cy0, Cy1, B, *dims)
    ws = workspace(nb, x0.device) if nb else None
    nvox = B * dims[0] * dims[1] * dims[2]
    flops = 2.0 * nvox * 125 * (C0 + C1) * (Cy0 + Cy1)
    nbytes = 2.0 * nvox * (C0 + C1 + Cy0 + Cy1) + 2.0 * 125 * (C0 + C1) * (Cy0 + Cy1)
    tag = "conv-bf16 k5 s1 %d^3x%d %d->%d" % (dims[2], B, C0 + C1, Cy0 + Cy1)
    acc = _ptr(acc_src) if acc_src is not None else (_ptr(y0) if accum else None)
    with _Timed(tag, flops, nbytes):
        if cin_real and _IN4["on"] and x1 is None and y1 is None and not accum and acc_src is None:
            check(L.vnet_conv_fwd_b16_padded(_ptr(x0), C0, int(cin_real), _ptr(wp), _ptr(bias), _ptr(y0), Cy0, B, *dims,
                                             _ptr(res), _ptr(stats), _ptr(ws), nb, _stream()), "vnet_conv_fwd_b16_padded")
            return
        check(L.vnet_conv_fwd_b16(_ptr(x0), C0, _ptr(x1), C1, _ptr(wp), _ptr(bias), _ptr(y0), Cy0, _ptr(y1), Cy1, B, *dims,
                                  acc, _ptr(res), _ptr(stats), _ptr(ws), nb, _stream()), "vnet_conv_fwd_b16")


def _wgrad5_b16_call(x0, x1, dy, dw, dims, cin_dw, owner=None):
    L = _lib.lib()
    B = x0.shape[0]
    C0, C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    Co = dy.shape[-1]
    nb = L.vnet_wgrad_bf16_ws_bytes(C0 + C1, Co, B, *dims)
    ws = _wgrad_workspace(dw, nb, False, owner)
    nvox = B * dims[0] * dims[1] * dims[2]
    flops = 2.0 * nvox * 125 * (C0 + C1) * Co
    nbytes = 2.0 * nvox * (C0 + C1 + Co) + 4.0 * 125 * (C0 + C1) * Co
    tag = _wgrad_tag(True, 5, 0, 1, dims[2], B, C0 + C1, Co)
    if (_DEFER["on"] and owner is not None and _GROUP["on"] and not _timed_tag(tag) and _LAUNCH_ON[0] is None
            and dims[0] * dims[1] * dims[2] <= _GROUP["max_voxels"] and not (cin_dw <= 4 and C0 == 8 and C1 == 0)):
        # a deep-level layer of a pass whose filter gradients nobody reads before it ends: launched together with the others when the
        # pass ends (vnet_conv_wgrad_b16_group); the tensors stay alive -- and unmodified, see _ConvFn.backward -- until then
        _DEFER["jobs"].append((x0, x1, dy, dw, ws, nb, int(cin_dw), B, tuple(dims), flops, nbytes, 5))
        _DEFER["dy_ptrs"].add(dy.data_ptr())
        _group_pinned(x0, x1, dy)
        return
    with _Timed(tag, flops, nbytes), _immediate_reduce(owner is None):
        check(L.vnet_conv_wgrad_b16(_ptr(x0), C0, _ptr(x1), C1, _ptr(dy), Co, _ptr(dw), int(cin_dw), B, *dims, _ptr(ws), nb, _stream()),
              "vnet_conv_wgrad_b16")


def _conv2_b16_call(up, x, wp, bias, y, dims_in, dims_out, accum=False, stats=None):
    L = _lib.lib()
    B, Cin, Cout = x.shape[0], x.shape[-1], y.shape[-1]
    nb = L.vnet_conv_ws_bytes(2, 0, 2, up, Cin, Cout, B, *dims_out)
    ws = workspace(nb, x.device) if nb else None
    nin, nout = B * dims_in[0] * dims_in[1] * dims_in[2], B * dims_out[0] * dims_out[1] * dims_out[2]
    flops = 2.0 * (nin if up else nout) * 8 * Cin * Cout
    nbytes = 2.0 * (nin * Cin + nout * Cout) + 4.0 * 8 * Cin * Cout
    tag = "conv-b16 k2 s2%s %d^3x%d %d->%d" % (" up" if up else "", dims_out[2], B, Cin, Cout)
    with _Timed(tag, flops, nbytes):
        check(L.vnet_conv2_fwd_b16(int(up), _ptr(x), Cin, _ptr(wp), _ptr(bias), _ptr(y), Cout, B, *dims_in, *dims_out,
                                   int(bool(accum)), _ptr(stats), _ptr(ws), nb, _stream()), "vnet_conv2_fwd_b16")


_DIRECT2 = {"on": True}          # (tests / A-B: False = always the generic kernels)


def _conv2_b16(down, x, w, bias, y, dims_fine, dims_coarse, Cf, Cc, accum=False, stats=None):
    """The 2^3 stride-2 pair (fp32 or bf16 tensors, by the dtype of x).  down: coarse y = conv(fine x); else fine y (+)= transposed conv(coarse x).
    w: the fp32 filter in TF layout -- [2,2,2,Cf,Cc] for BOTH layers2.down_convolution (Cin = Cf) and layers2.up_convolution
    (filter [k,k,k,Cout = Cf,Cin = Cc]).  Levels 1-2 of the V-Net (Cf 16 / 32) take the LDS-free direct kernels
    (csrc/conv2_b16.hip: one 16-byte load = one MFMA operand), other widths the generic fp32-MFMA kernels on packed weights."""
    L = _lib.lib()
    B = x.shape[0]
    f32 = x.dtype == torch.float32
    if _DIRECT2["on"] and L.vnet_conv2_direct_ok(Cf, Cc) and w.data_ptr() % 16 == 0:
        nf = B * dims_fine[0] * dims_fine[1] * dims_fine[2]
        nc = B * dims_coarse[0] * dims_coarse[1] * dims_coarse[2]
        tag = "conv%s k2 s2%s %d^3x%d %d->%d" % ("" if f32 else "-b16", "" if down else " up", (dims_coarse if down else dims_fine)[2], B,
                                                 Cf if down else Cc, Cc if down else Cf)
        esz = 4.0 if f32 else 2.0
        with _Timed(tag, 2.0 * nc * 8 * Cf * Cc, esz * (nf * Cf + nc * Cc) + 4.0 * 8 * Cf * Cc):
            fn, what = (L.vnet_conv2_direct_f32, "vnet_conv2_direct_f32") if f32 else (L.vnet_conv2_direct_b16, "vnet_conv2_direct_b16")
            check(fn(int(bool(down)), _ptr(x), _ptr(y), _ptr(w), _ptr(bias), Cf, Cc, B, *dims_fine, *dims_coarse,
                     int(bool(accum)), _ptr(stats), _stream()), what)
        return
    if f32:             # the generic fp32 MFMA kernels on packed filters (any width)
        if down:
            _conv_call(2, 2, 0, x, None, packed_weights(w, PACK_FWD, 8, Cf, Cc), bias, y, None, dims_fine, dims_coarse, accum=accum, stats=stats)
        else:
            _conv_call(2, 2, 1, x, None, packed_weights(w, PACK_UP, 8, Cc, Cf), bias, y, None, dims_coarse, dims_fine, accum=accum)
        return
    if down:
        _conv2_b16_call(0, x, packed_weights(w, PACK_FWD | PACK_ROUND16, 8, Cf, Cc), bias, y, dims_fine, dims_coarse, accum=accum, stats=stats)
    else:
        _conv2_b16_call(1, x, packed_weights(w, PACK_UP | PACK_ROUND16, 8, Cc, Cf), bias, y, dims_coarse, dims_fine, accum=accum)


def _wgrad2_b16_call(xfine, dycoarse, dw, dims_fine, dims_coarse, owner=None):
    L = _lib.lib()
    immediate = owner is None
    B, Cin, Co = xfine.shape[0], xfine.shape[-1], dycoarse.shape[-1]
    nb = L.vnet_wgrad_ws_bytes(2, 0, 2, Cin, Co, B, *dims_coarse)
    ws = _wgrad_workspace(dw, nb, immediate, owner)
    nout = B * dims_coarse[0] * dims_coarse[1] * dims_coarse[2]
    tag = "wgrad-b16 k2 s2 %d^3x%d %d->%d" % (dims_coarse[2], B, Cin, Co)
    if (_DEFER["on"] and not immediate and _GROUP["on"] and _GROUP["k2"] and not _timed_tag(tag) and _LAUNCH_ON[0] is None
            and tuple(dims_coarse) == tuple((d + 1) // 2 for d in dims_fine)
            and dims_fine[0] * dims_fine[1] * dims_fine[2] <= _GROUP["max_voxels"]):
        # joins the grouped launch of the pass's filter gradients (ks = 2: x = the fine tensor, dy = the coarse one)
        _DEFER["jobs"].append((xfine, None, dycoarse, dw, ws, nb, int(Cin), B, tuple(dims_fine),
                               2.0 * nout * 8 * Cin * Co, 2.0 * nout * (8 * Cin + Co) + 4.0 * 8 * Cin * Co, 2))
        _DEFER["dy_ptrs"].add(dycoarse.data_ptr())
        _group_pinned(xfine, None, dycoarse)
        return
    with _Timed(tag, 2.0 * nout * 8 * Cin * Co, 2.0 * nout * (8 * Cin + Co) + 4.0 * 8 * Cin * Co), _immediate_reduce(immediate):
        check(L.vnet_conv2_wgrad_b16(_ptr(xfine), Cin, _ptr(dycoarse), Co, _ptr(dw), B, *dims_fine, *dims_coarse, _ptr(ws), nb, _stream()),
              "vnet_conv2_wgrad_b16")


def cast_input(img):
    """bf16-storage mode: the fp32 network input [.., Cin] as a bf16 tensor [.., Cin padded to a multiple of 8] (zero channels);
    the filters' packed images are zero-padded to 16 input channels anyway (vnet_cast_bf16).  Not differentiable: the reference's
    input is a placeholder (model.py:262)."""
    _need_gpu(img, "cast_input")
    img = img.contiguous()
    C = int(img.shape[-1])
    Cp = -(-C // 8) * 8
    y = torch.empty(img.shape[:-1] + (Cp,), dtype=torch.bfloat16, device=img.device)
    check(_lib.lib().vnet_cast_bf16(_ptr(img), _ptr(y), img.numel() // C, C, Cp, _stream()), "vnet_cast_bf16")
    return y


def colsum16(x16, C, out):
    """Per-channel fp32 sum of a bf16 channels-last tensor (bias gradients outside the networks' closed form): written by a kernel
    on the launch stream straight into `out` (a raw-pointer write like colsum(): no copy_ on torch's current stream, no version
    bump of a gradient-sink view)."""
    L = _lib.lib()
    M = x16.numel() // C
    nb = L.vnet_colsum_b16_ws_bytes(C)
    ws = workspace(nb, x16.device)
    check(L.vnet_colsum_b16(_ptr(x16), _ptr(out), M, C, _ptr(ws), nb, _stream()), "vnet_colsum_b16")
    return out


# ---- deferred reduces of the filter-gradient slabs (include/vnet_hip.h: vnet_wgrad_defer / vnet_wgrad_flush) ------------------
# Inside `deferred_wgrad_reduce()` the filter-gradient launches leave their partial slabs in a per-layer buffer and ONE batched
# launch reduces all of them when the context ends (26 reduce launches of ~7 us per V-Net step otherwise).  Only for a backward
# pass whose filter gradients nobody reads before it ends (model.image2label: not the eager data-parallel step, whose bucket
# all-reduces start from the gradient hooks).
_DEFER = {"on": False, "jobs": [], "dy_ptrs": set(), "stream": None}
# grouped launch of the 5^3 filter gradients of a deferring pass (layers up to 128^3 voxels; measured: 32^3 and below -0.23 ms,
# all levels -0.33 ms per C5 step) (bf16 storage; include/vnet_hip.h:
# vnet_conv_wgrad_b16_group).  VNET_WGRAD_GROUP=0: every layer launches its own kernel as it did through round 3.
_GROUP = {"on": _os.environ.get("VNET_WGRAD_GROUP", "1") != "0", "max_voxels": 128 ** 3, "max_bytes": 4 << 30,
          "k2": True}
# (k2: the 2^3 stride-2 filter gradients join too -- -0.05 ms per C5 step.  Round 4 also built the group for fp32 tensors and let the
#  zero-padded network input join; both measured no gain -- DESIGN 4.5 -- and were removed in round 5)


WGRAD_GROUP_TAG = "wgrad-group"


def set_wgrad_group(on):
    _GROUP["on"] = bool(on)


def _group_pinned(*tensors):
    """Byte budget of the grouped launch: the activations and gradients of every layer that joins stay alive until the
    group is launched, where autograd used to release them layer by layer; the footprint grows with the batch size.  Once the
    tensors waiting exceed `_GROUP["max_bytes"]` the layers collected so far are launched mid-pass (their slabs' reduces still join
    the one batched flush) and their tensors are let go.  An upper bound: a tensor two layers share is counted twice."""
    _DEFER["pinned"] = _DEFER.get("pinned", 0) + sum(t.numel() * t.element_size() for t in tensors if t is not None)
    if _DEFER["pinned"] > _GROUP["max_bytes"]:
        _flush_wgrad_group()


def _flush_wgrad_group(launch=True):
    """Launch the collected filter gradients (the reduces of their slabs join the deferred queue) and let go of their tensors."""
    jobs, _DEFER["jobs"] = _DEFER["jobs"], []
    _DEFER["dy_ptrs"] = set()
    _DEFER["pinned"] = 0
    if not jobs or not launch:
        return
    _launch_wgrad_group(jobs, "vnet_conv_wgrad_b16_group")


def _launch_wgrad_group(jobs, entry):
    L = _lib.lib()
    arr = (_lib.WgradJob * len(jobs))()
    for k, (x0, x1, dy, dw, ws, nb, cin_dw, B, dims, _fl, _by, ks) in enumerate(jobs):
        j = arr[k]
        j.x0, j.x1, j.dy, j.dw, j.ws, j.ws_bytes = _ptr(x0), _ptr(x1), _ptr(dy), _ptr(dw), _ptr(ws), int(nb)
        j.C0, j.C1, j.Cout, j.Cin_dw = int(x0.shape[-1]), (int(x1.shape[-1]) if x1 is not None else 0), int(dy.shape[-1]), cin_dw
        j.B, j.D, j.H, j.W, j.ks = int(B), int(dims[0]), int(dims[1]), int(dims[2]), ks
    # (a layer that is being timed on its own -- profile_start(only=...) with its tag -- has not joined; "wgrad-group" times this launch)
    with _Timed(WGRAD_GROUP_TAG, sum(j[9] for j in jobs), sum(j[10] for j in jobs)):
        check(getattr(L, entry)(ctypes.addressof(arr), len(jobs), _stream()), entry)


@contextlib.contextmanager
def deferred_wgrad_reduce(on=True):
    if not on or _DEFER["on"]:
        yield
        return
    L = _lib.lib()
    st = _stream()                            # the library's queues are per stream: this pass defers on the stream it launches on
    _DEFER["on"] = True
    _DEFER["stream"] = st
    L.vnet_wgrad_defer(1, st)
    try:
        yield
    except BaseException:
        # the pass failed (e.g. an invalidated stream capture): drain the queue without raising a SECOND error from here, so that
        # the caller sees the original one (model._train_step_graph turns a refused capture into eager steps)
        _DEFER["on"] = False
        _flush_wgrad_group(launch=False)
        L.vnet_wgrad_defer(0, st)
        L.vnet_wgrad_flush(st)
        raise
    else:
        try:
            _flush_wgrad_group()                  # (still deferring: the reduces of its slabs join the one batched launch below)
        finally:
            _DEFER["on"] = False
            L.vnet_wgrad_defer(0, st)
        check(L.vnet_wgrad_flush(st), "vnet_wgrad_flush")


def _wgrad_workspace(dw, nbytes, immediate, owner):
    """Scratch for the partial slabs: the shared workspace, or (deferred reduce) a buffer of this layer's own, kept on the
    parameter's gradient sink so that it lives exactly as long as the parameter does (a captured step graph holds its address)."""
    if not _DEFER["on"] or immediate or owner is None:
        return workspace(nbytes, dw.device)
    if owner.ws is None or owner.ws.numel() < nbytes or owner.ws.device != dw.device:
        if owner.ws is not None:
            _WS_RETIRED.append(owner.ws)          # (a captured graph may hold the old address: keep it alive, see workspace())
        owner.ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dw.device)
    return owner.ws


@contextlib.contextmanager
def _immediate_reduce(immediate):
    """A filter gradient that is consumed right away (the fused input block's G): reduce it now even inside a deferring pass."""
    if not (immediate and _DEFER["on"]):
        yield
        return
    L = _lib.lib()
    st = _DEFER.get("stream")
    L.vnet_wgrad_defer(0, st)
    try:
        yield
    finally:
        L.vnet_wgrad_defer(1, st)


def _wgrad_x3_ok(C0, C1, Co, B, dims):
    if not _COMPUTE.get("split3"):
        return False
    if _X3["force"]:
        return C0 % 16 == 0 and C1 % 16 == 0 and Co % 16 == 0
    return _lib.lib().vnet_wgrad_x3_ok(C0, C1, Co, B, *dims) == 1


def _wgrad_x3_call(x0, x1, dy, dw, dims, owner=None):
    """Filter gradient of the 5^3 stride-1 conv, fp32 tensors, products from three-way split bf16 operands (vnet_conv_wgrad_x3)."""
    L = _lib.lib()
    B = x0.shape[0]
    C0, C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    Co = dy.shape[-1]
    nb = L.vnet_wgrad_x3_ws_bytes(C0 + C1, Co, B, *dims)
    ws = _wgrad_workspace(dw, nb, owner is None, owner)
    nvox = B * dims[0] * dims[1] * dims[2]
    flops = 2.0 * nvox * 125 * (C0 + C1) * Co
    nbytes = 4.0 * (nvox * (C0 + C1 + Co) + 125 * (C0 + C1) * Co)
    tag = "wgrad-x3 k5 s1 %d^3x%d %d->%d" % (dims[2], B, C0 + C1, Co)
    with _Timed(tag, flops, nbytes), _immediate_reduce(owner is None):
        check(L.vnet_conv_wgrad_x3(_ptr(x0), C0, _ptr(x1), C1, _ptr(dy), Co, _ptr(dw), B, *dims, _ptr(ws), nb, _stream()),
              "vnet_conv_wgrad_x3")


def _wgrad_call(ks, stride, x0, x1, dy, dw, dims_in, dims_out, kx=0, immediate=False, owner=None):
    L = _lib.lib()
    immediate = immediate or owner is None          # no sink to keep the slabs on: reduce on the spot
    B = x0.shape[0]
    C0, C1 = x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    Co = dy.shape[-1]
    nb = L.vnet_wgrad_ws_bytes(ks, kx, stride, C0 + C1, Co, B, *dims_out)
    ws = _wgrad_workspace(dw, nb, immediate, owner)
    nin, nout = B * dims_in[0] * dims_in[1] * dims_in[2], B * dims_out[0] * dims_out[1] * dims_out[2]
    taps = ks * ks * (kx or ks)
    flops = 2.0 * nout * taps * (C0 + C1) * Co
    nbytes = 4.0 * (nin * (C0 + C1) + nout * Co + taps * (C0 + C1) * Co)
    tag = _wgrad_tag(False, ks, kx, stride, dims_out[2], B, C0 + C1, Co)
    with _Timed(tag, flops, nbytes), _immediate_reduce(immediate):
        check(L.vnet_conv_wgrad(ks, kx, stride, _ptr(x0), C0, _ptr(x1), C1, _ptr(dy), Co, _ptr(dw),
                                B, *dims_in, *dims_out, _ptr(ws), nb, _stream()), "vnet_conv_wgrad")


def colsum(x2d_like, C, out=None):
    """Per-channel sum over all leading axes of a channels-last tensor."""
    L = _lib.lib()
    M = x2d_like.numel() // C
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x2d_like.device)
    nb = L.vnet_colsum_ws_bytes(C)
    ws = workspace(nb, x2d_like.device)
    check(L.vnet_colsum(_ptr(x2d_like), _ptr(out), M, C, _ptr(ws), nb, _stream()), "vnet_colsum")
    return out


# ---- tensors with two consumers: the second gradient is accumulated by the kernel that produces it ---------------------
# The skip connection (networks.py:276 -> 325: block output feeds the down convolution AND the decoder's concat) and the
# residual blocks (networks.py:314-318: block input feeds conv_1 AND the add in front of the last batch-norm) give autodiff
# two gradient contributions per tensor, which it sums with an add kernel (tf.add_n / torch's accumulation: 8 full-tensor
# adds per step, 134 MB each at level 1).  `fork` hands the two consumers separate views that share a slot; the consumer
# whose backward runs FIRST leaves its gradient tensor in the slot, the second one runs its backward-data kernel in
# accumulate mode (y += ...) on that tensor and reports no gradient of its own, so the sum costs no extra pass.
class _GradSlot(object):
    __slots__ = ("first", "total")

    def __init__(self):
        self.first = None
        self.total = None        # the SUM of both gradients, written out of place by the second consumer (see _ConvFn.backward)


class _ForkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.set_materialize_grads(False)
        ctx.slot = slot
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        total, ctx.slot.total = ctx.slot.total, None
        ctx.slot.first = None    # drop the slot's reference: a leaf behind this node can then take the gradient without a copy
        if total is not None:
            return total, None   # one consumer's kernel already added the other's gradient (out of place)
        if gb is None:
            return ga, None
        if ga is None:
            return gb, None
        return ga + gb, None     # both consumers reported a gradient (accumulation not possible there)


def fork(x):
    """Two handles of `x` for its two consumers (see above); plain (x, x) when no gradient is being recorded."""
    if _meta(x) or not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    slot = _GradSlot()
    a, b = _ForkFn.apply(x, slot)
    a._vnet_slot = b._vnet_slot = slot
    return a, b


def cut(x, registry):
    """Backward cut point (data-parallel step graphs): returns a LEAF that shares x's storage; `registry` collects
    (x, leaf).  A first backward pass stops at the leaves (their .grad holds what arrived), a second one continues from the
    recorded tensors: torch.autograd.backward([x...], [leaf.grad...]).  model.image2label uses it to end the first gradients
    graph where 81 % of the gradient bytes exist, so that their all-reduce travels under the rest of backward."""
    if _meta(x) or not (torch.is_grad_enabled() and x.requires_grad):
        return x
    leaf = x.detach().requires_grad_(True)
    registry.append((x, leaf))
    return leaf


def _slot_target(slot, dy, shape):
    """The tensor an accumulating backward-data kernel may add into: the other consumer's gradient, unless the parameter-
    gradient stream is on (its filter-gradient launches may still be reading that tensor) or it IS this kernel's input."""
    if slot is None or slot.first is None or _PG["on"]:
        return None
    t = slot.first
    if t.data_ptr() == dy.data_ptr() or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        return None
    return t


# ---- conv bias in front of a batch-norm ----------------------------------------------------------------------------------
# Every convolution of the V-Net (both wirings) feeds a train-mode batch-norm (networks.py:259...361; decoder chains
# included), and a batch-norm's output does not change when a per-channel constant is added to its input -- the batch mean
# absorbs it.  So dLoss/dbias == 0 IDENTICALLY for every conv bias of the network; what the reference's autodiff computes
# there (sum of dy over voxels, with dy the batch-norm's input gradient) is the floating-point residue of a sum that is zero
# in exact arithmetic (the fp64 oracle gets ~1e-12).  Inside this context the convolutions take the closed form: the bias
# gradient is left at exactly 0 in the flat gradient buffer and the 29 column-sum + 29 finalize launches per step are not
# made.  Stand-alone layers2.convolution (outside the networks) keeps the generic column sum.
_FUSE = {"zero_bias_grad": False, "bn_stats": True,
         "bn_stats_fp32_direct": True,
         # round 6: the single-modality input block on the vector pipe straight from the image (csrc/input_block.hip:
         # input_conv_direct_kernel / input_wgrad_direct_kernel); False = rounds 1-5: x-im2col tensor + 5x5x1 fp32-MFMA kernels (A/B runs)
         "input_direct": _os.environ.get("VNET_INPUT_DIRECT", "1") != "0"}


def set_epilogue_bn_stats(on, fp32_direct=None):
    """Switch for the batch-norm statistics in the convolution epilogues (on by default; off = separate statistics pass).
    fp32_direct: also in the epilogue of the non-split fp32 MFMA kernels (on by default; worth 0.07 ms per 128^3 step)."""
    _FUSE["bn_stats"] = bool(on)
    if fp32_direct is not None:
        _FUSE["bn_stats_fp32_direct"] = bool(fp32_direct)


@contextlib.contextmanager
def zero_bias_gradients(on=True):
    prev = _FUSE["zero_bias_grad"]
    _FUSE["zero_bias_grad"] = bool(on)
    try:
        yield
    finally:
        _FUSE["zero_bias_grad"] = prev


class _ConvFn(torch.autograd.Function):
    """conv (ks=5,s=1 | ks=2,s=2) or 2^3 transposed conv (up) + bias, two-source input."""

    @staticmethod
    def forward(ctx, x0, x1, w, b, ks, stride, up, out_spatial, stats=None, res=None):
        slot0, slot1 = getattr(x0, "_vnet_slot", None), getattr(x1, "_vnet_slot", None)
        x0 = x0.contiguous()
        x1 = x1.contiguous() if x1 is not None else None
        B, Di, Hi, Wi, C0 = x0.shape
        C1 = x1.shape[-1] if x1 is not None else 0
        if up:
            O, I = w.shape[-2], w.shape[-1]
            dims_out = tuple(int(v) for v in out_spatial)
            wp = None
        else:
            I, O = w.shape[-2], w.shape[-1]
            dims_out = (_same_out(Di, stride), _same_out(Hi, stride), _same_out(Wi, stride))
            wp = None
        b16 = _is16(x0)            # bf16-storage mode: bf16 tensors in and out (the network input may be zero-padded to 8 channels)
        if I != C0 + C1 and not (b16 and x1 is None and C0 == -(-I // 8) * 8):
            raise VnetHipError("conv: filter expects %d input channels, got %d" % (I, C0 + C1))
        if b16 and (_is16(x1) != (x1 is not None) or (res is not None and not _is16(res))):
            raise VnetHipError("conv: bf16 and float32 tensors mixed")
        y = torch.empty((B,) + dims_out + (O,), dtype=torch.bfloat16 if b16 else torch.float32, device=x0.device)
        bf16 = (not up) and ks == 5 and stride == 1 and b16
        if res is not None:
            res = res.contiguous()
        if b16:
            if x1 is not None and (up or ks != 5):
                raise VnetHipError("conv: the two-source form exists for the 5^3 convolution only")
            if up:          # w [2,2,2,O,I]: fine channels O, coarse channels I
                _conv2_b16(False, x0, w, b, y, dims_out, (Di, Hi, Wi), O, I)
            elif ks == 2:   # w [2,2,2,I,O]: fine channels I, coarse channels O
                _conv2_b16(True, x0, w, b, y, (Di, Hi, Wi), dims_out, I, O, stats=stats)
            else:
                _conv5_b16_call(x0, x1, packed_weights(w, PACK_FWD_BF16, 125, I, O), b, y, None, dims_out, stats=stats, res=res,
                                cin_real=(I if x1 is None and I < x0.shape[-1] else 0))
        elif up and x1 is None:
            _conv2_b16(False, x0, w, b, y, dims_out, (Di, Hi, Wi), O, I)
        elif ks == 2 and stride == 2 and x1 is None and res is None:
            _conv2_b16(True, x0, w, b, y, (Di, Hi, Wi), dims_out, I, O, stats=stats)
        elif ks == 5 and stride == 1 and not up and _x3_ok(C0, C1, O, 0, B, dims_out):
            _conv_x3_call(x0, x1, packed_weights(w, PACK_FWD_X3, 125, I, O), b, y, None, dims_out, stats=stats, res=res)
        else:
            if wp is None:
                wp = packed_weights(w, PACK_UP, 8, I, O) if up else packed_weights(w, PACK_FWD, ks ** 3, I, O)
            _conv_call(ks, stride, 1 if up else 0, x0, x1, wp, b, y, None, (Di, Hi, Wi), dims_out, stats=stats, res=res)
        ctx.save_for_backward(x0, x1, w)
        ctx.params = (w, b)
        ctx.cfg = (ks, stride, up, (Di, Hi, Wi), dims_out, C0, C1, I, O)
        ctx.bf16 = bf16
        ctx.b16 = b16
        ctx.bias_zero = _FUSE["zero_bias_grad"] and b is not None
        ctx.slots = (slot0, slot1)
        ctx.opsctx = current_context()
        return y

    @staticmethod
    def backward(ctx, dy):
        # the backward pass runs in the OpsContext of the forward pass (a user-driven loss.backward() outside
        # model.in_context would otherwise pick the default context's compute mode, stream and pack registry)
        with context(ctx.opsctx):
            return _ConvFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x0, x1, w = ctx.saved_tensors
        ks, stride, up, din, dout, C0, C1, I, O = ctx.cfg
        dy = dy.contiguous()
        B = x0.shape[0]
        dev = x0.device
        wref, bref = ctx.params
        db = dw = None
        sb = sw = None
        bias_zero = False
        if ctx.needs_input_grad[3]:
            if ctx.bias_zero and getattr(bref, "_vnet_sink", None) is not None and not bref._vnet_sink.written:
                bias_zero, sb = True, bref._vnet_sink      # closed form: the flat gradient buffer already holds the exact 0
            else:
                db, sb = _grad_out(bref)
        if ctx.needs_input_grad[2]:
            dw, sw = _grad_out(wref)
        side = param_grad_stream(dev)
        if side is not None and ((db is not None and sb is None) or (dw is not None and sw is None)):
            side = None                                  # a gradient autograd has to hand on: stay on the main stream
        if side is not None and dw is not None and _PROFILE["on"]:
            wtag = (_wgrad_tag(False, 2, 0, 2, din[2], B, O, I) if up else
                    _wgrad_tag(ctx.bf16, ks, 0, stride, dout[2], B, C0 + C1, O))
            if _timed_tag(wtag):
                side = None                              # a launch that is being timed runs alone (bench.py roofline)
        if side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)                       # dy is complete on the main stream
            for t in (x0, x1, dy):
                if t is not None:
                    t.record_stream(side)                # keep the allocator from recycling them under the side stream
            # dy of a residual block is ALSO the gradient of the block input (one tensor, two autograd edges); once this
            # node has returned autograd holds the last reference and accumulates the next gradient INTO it in place
            # (input_buffer.cpp: use_count == 1), on the main stream, while the filter gradient may still be reading it.
            # A live reference until the join makes that accumulation out of place.
            if not _PG["used"]:
                # first side-stream launch of this backward pass: join when the pass ends, whoever drives it (a user
                # loop calling loss.backward() directly would otherwise keep every dy alive and could read filter
                # gradients before the side stream has written them)
                torch.autograd.Variable._execution_engine.queue_callback(join_param_grad_stream)
            _PG["keep"].append(dy)
            _PG["used"].add(dev)
            if _PG["test_delay"]:
                with torch.cuda.stream(side):
                    torch.cuda._sleep(int(_PG["test_delay"]))    # tests: let the side stream lag far behind
            _LAUNCH_ON[0] = side.cuda_stream             # launches (and their scratch buffer) go to the side stream
        b16 = ctx.b16
        try:
            if db is not None:
                if b16:
                    colsum16(dy, O, db)
                else:
                    colsum(dy, O, out=db)
            if dw is not None and b16:
                if up:
                    _wgrad2_b16_call(dy, x0, dw, dout, din, owner=sw)
                elif stride == 2:
                    _wgrad2_b16_call(x0, dy, dw, din, dout, owner=sw)
                else:
                    _wgrad5_b16_call(x0, x1, dy, dw, din, I, owner=sw)
            elif dw is not None:
                if up:      # dw[a][o][ci] = sum_i dy[2i+a][o] * x[i][ci]  == filter grad of the 2^3 down conv (fine -> coarse)
                    _wgrad_call(2, 2, dy, None, x0, dw, dout, din, owner=sw)
                elif ks == 5 and stride == 1 and _wgrad_x3_ok(C0, C1, O, B, din):
                    _wgrad_x3_call(x0, x1, dy, dw, din, owner=sw)
                else:
                    _wgrad_call(ks, stride, x0, x1, dy, dw, din, dout, owner=sw)
        finally:
            _LAUNCH_ON[0] = None
        dx0 = dx1 = None
        r0 = r1 = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            slot0, slot1 = ctx.slots
            # x0 has a second consumer whose gradient exists already: add this one into it (single-source convs only)
            acc = _slot_target(slot0, dy, x0.shape) if x1 is None else None
            # one-convolution residual block: the other gradient of x0 IS this node's dy (batch-norm's ds serves the conv
            # output and the residual) -- no in-place sum, but the bf16 kernels can add it on the way out: dx0 = conv(dy) + dy
            oop = None
            if (acc is None and x1 is None and ctx.bf16 and slot0 is not None and slot0.first is not None
                    and slot0.first.data_ptr() == dy.data_ptr() and tuple(dy.shape) == tuple(x0.shape)
                    and b16 and C0 % 8 == 0):
                oop = dy
            if acc is not None and b16 and ks == 5 and stride == 1 and not up and acc.data_ptr() in _DEFER["dy_ptrs"]:
                # the other gradient of x0 is ALSO the dy of a filter gradient that waits for the grouped launch (a residual block's
                # ds serves its last convolution and the block input): add it on the way out into a fresh tensor instead of in place
                # -- the same arithmetic, RNE(accumulator + stored gradient), so the bits do not change
                oop, acc = acc, None
            dx0 = acc if acc is not None else torch.empty_like(x0)
            dx1 = torch.empty_like(x1) if x1 is not None else None
            accum = acc is not None
            if b16 and up:      # backward-data of the transposed conv = the 2^3 stride-2 conv with the same filter
                _conv2_b16(True, dy, w, None, dx0, dout, din, O, I, accum=accum)
            elif b16 and stride == 2:
                _conv2_b16(False, dy, w, None, dx0, din, dout, I, O, accum=accum)
            elif b16:
                _conv5_b16_call(dy, None, packed_weights(w, PACK_BWD_BF16, 125, I, O), None, dx0, dx1, din, accum=accum, acc_src=oop)
                if oop is not None:
                    slot0.total = dx0
            elif up:        # backward-data of the transposed conv = the 2^3 stride-2 conv with the same filter
                _conv2_b16(True, dy, w, None, dx0, dout, din, O, I, accum=accum)
            elif stride == 2:   # backward-data of the down conv = the 2^3 transposed conv with the same filter
                _conv2_b16(False, dy, w, None, dx0, din, dout, I, O, accum=accum)
            elif ks == 5 and _x3_ok(O, 0, C0, C1, B, din):
                _conv_x3_call(dy, None, packed_weights(w, PACK_BWD_X3, 125, I, O), None, dx0, dx1, din, accum=accum)
            else:
                wp = packed_weights(w, PACK_BWD, ks ** 3, I, O)
                _conv_call(ks, 1, 0, dy, None, wp, None, dx0, dx1, dout, din, accum=accum)
            r0, r1 = (None if accum else dx0), dx1
            if not accum and slot0 is not None and slot0.first is None:
                slot0.first = dx0                      # first of the two gradients of a forked tensor
            if slot1 is not None and slot1.first is None:
                slot1.first = dx1
        gb = _grad_ret(db, sb) if (db is not None or bias_zero) else None
        return r0, r1, _grad_ret(dw, sw), gb, None, None, None, None, None, None


def _meta(*ts):
    return any(t is not None and t.device.type == "meta" for t in ts)


class _InputConvFn(torch.autograd.Function):
    """conv5^3(BN(tile(img))) + b for a 1-channel image without the 16x redundant work (csrc/input_block.hip):
    forward = 5x5x1 conv over the x-im2col of (img, inside-indicator) with BN-folded filters; backward needs only
    the 2-channel filter gradient G -- dw, and the conv-path parts of the input BN's dgamma/dbeta follow from it."""

    @staticmethod
    def forward(ctx, img, gamma, beta, mean, invstd, w, b, stats=None, res=None):
        L = _lib.lib()
        img = img.contiguous()
        if res is not None:
            res = res.contiguous()
        B, D, H, W, _ = img.shape
        C, O = w.shape[-2], w.shape[-1]
        dev = img.device
        wv = torch.empty((25, 16, O), dtype=torch.float32, device=dev)
        check(L.vnet_input_conv_fold(_ptr(w), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(wv), C, O, _stream()),
              "vnet_input_conv_fold")
        y = torch.empty((B, D, H, W, O), dtype=torch.float32, device=dev)
        direct = _input_direct_ok(O, B, D, H, W)
        if direct:
            # packed fp32 FMAs straight from the image: no im2col tensor, no filter repack
            xv = img
            border = torch.empty(9 * 26 * O, dtype=torch.float32, device=dev)      # wbc [9][25][O] | cbc [9][O]
            check(L.vnet_input_conv_fold_border(_ptr(wv), O, _ptr(border), _ptr(border[9 * 25 * O:]), _stream()), "vnet_input_conv_fold_border")
            with _Timed("input-direct %d^3x%d 1->%d" % (W, B, O), 2.0 * B * D * H * W * 125 * O, 4.0 * B * D * H * W * (1 + O)):
                check(L.vnet_input_conv_direct_fwd(_ptr(img), _ptr(wv), _ptr(border), _ptr(border[9 * 25 * O:]), _ptr(b), _ptr(res), _ptr(y),
                                                   _ptr(stats), O, B, D, H, W, _stream()), "vnet_input_conv_direct_fwd")
        else:
            xv = torch.empty((B, D, H, W, 16), dtype=torch.float32, device=dev)
            check(L.vnet_tile_im2col_x(_ptr(img), _ptr(xv), B, D, H, W, _stream()), "vnet_tile_im2col_x")
            wp = torch.empty(L.vnet_packed_weight_floats(PACK_FWD, 25, 16, O), dtype=torch.float32, device=dev)
            check(L.vnet_pack_weights(PACK_FWD, _ptr(wv), _ptr(wp), 25, 16, O, _stream()), "vnet_pack_weights")
            _conv_call(5, 1, 0, xv, None, wp, b, y, None, (D, H, W), (D, H, W), kx=1, stats=stats, res=res)
        ctx.direct = direct
        ctx.save_for_backward(xv, gamma, beta, mean, invstd, w)
        ctx.params = (w, b)
        ctx.gb = (gamma, beta)
        ctx.bias_zero = _FUSE["zero_bias_grad"]
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        xv, gamma, beta, mean, invstd, w = ctx.saved_tensors
        wref, bref = ctx.params
        dy = dy.contiguous()
        B, D, H, W = xv.shape[:4]
        C, O = w.shape[-2], w.shape[-1]
        dev = dy.device
        if ctx.bias_zero and getattr(bref, "_vnet_sink", None) is not None and not bref._vnet_sink.written:
            db, sb = None, bref._vnet_sink               # closed form (see zero_bias_gradients)
        else:
            db, sb = _grad_out(bref)
            colsum(dy, O, out=db)
        G = torch.empty((25, 16, O), dtype=torch.float32, device=dev)
        if ctx.direct:                                   # xv IS the image here
            nb = L.vnet_input_wgrad_direct_slabs(B, D, H, W) * 25 * 16 * O * 4
            ws = workspace(nb, dev)
            with _Timed("input-wgrad-direct %d^3x%d 1->%d" % (W, B, O), 2.0 * B * D * H * W * 125 * O, 4.0 * B * D * H * W * (1 + O)):
                check(L.vnet_input_wgrad_direct(_ptr(xv), _ptr(dy), _ptr(G), O, B, D, H, W, _ptr(ws), nb, _stream()), "vnet_input_wgrad_direct")
        else:
            _wgrad_call(5, 1, xv, None, dy, G, (D, H, W), (D, H, W), kx=1, immediate=True)     # G is folded right below
        dw, sw = _grad_out(wref)
        gpar, bpar = ctx.gb
        gs, bs = getattr(gpar, "_vnet_sink", None), getattr(bpar, "_vnet_sink", None)
        if (sw is not None and gs is not None and bs is not None and not gs.written and not bs.written
                and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]):
            # gamma / beta of the input batch-norm get two contributions: this one (through the folded filter) and the
            # batch-norm's own.  The batch-norm's backward runs later in this pass and WRITES its part into the flat buffer;
            # this kernel is deferred until then and ADDS its part (accumulate = 1) -- no autodiff add kernels, one launch.
            def deferred(accumulate, L=L, G=G, w=w, gamma=gamma, beta=beta, mean=mean, invstd=invstd, dw=dw, sw=sw, gs=gs, bs=bs):
                check(L.vnet_input_conv_grads(_ptr(G), _ptr(w), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(dw),
                                              _ptr(gs.view), _ptr(bs.view), C, O, int(accumulate), _stream()), "vnet_input_conv_grads")
                _grad_ret(dw, sw)
                if not accumulate:          # the batch-norm's backward never came: these sinks are complete now
                    _grad_ret(gs.view, gs)
                    _grad_ret(bs.view, bs)
            gpar._vnet_deferred = deferred

            def flush(gpar=gpar):
                th = getattr(gpar, "_vnet_deferred", None)
                if th is not None:
                    del gpar._vnet_deferred
                    th(0)
            torch.autograd.Variable._execution_engine.queue_callback(flush)
            return None, None, None, None, None, None, _grad_ret(db, sb), None, None
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        check(L.vnet_input_conv_grads(_ptr(G), _ptr(w), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(dw),
                                      _ptr(dgamma), _ptr(dbeta), C, O, 0, _stream()), "vnet_input_conv_grads")
        return None, dgamma, dbeta, None, None, _grad_ret(dw, sw), _grad_ret(db, sb), None, None


def _input_direct_ok(O, B, D, H, W):
    return _FUSE["input_direct"] and _lib.lib().vnet_input_conv_direct_ok(O, B, D, H, W) == 1


def input_conv(img, gamma, beta, mean, invstd, w, b, bn_stats=False, bn_residual=None):
    """convolution(BN(tf.tile(img)), [5,5,5,C,C]) for a 1-channel `img` (networks.py:254-259 + 316)."""
    if _meta(img):
        return torch.empty(img.shape[:-1] + (w.shape[-1],), device="meta")
    _need_gpu(img, "input_conv")
    stats = None
    if bn_stats and _FUSE["bn_stats"] and _FUSE["bn_stats_fp32_direct"] and _SYNC_BN is None:
        B, D, H, W, _ = img.shape
        if _input_direct_ok(w.shape[-1], B, D, H, W):
            rows = _lib.lib().vnet_input_conv_direct_stats_rows(B, D, H, W)
        else:
            rows = _lib.lib().vnet_conv_stats_rows(5, 1, 1, 0, 16, w.shape[-1], 0, B, D, H, W)
        if rows > 0:
            stats = torch.empty((rows, 2 * w.shape[-1]), dtype=torch.float32, device=img.device)
    if stats is None:
        return _InputConvFn.apply(img, gamma, beta, mean, invstd, w, b)
    y = _InputConvFn.apply(img, gamma, beta, mean, invstd, w, b, stats, bn_residual)
    y._vnet_stats = _EpilogueStats(stats, stats.shape[0], bn_residual)
    return y


class _EpilogueStats(object):
    """Partial batch-norm sums a convolution wrote in its epilogue for its output y: rows x [sum(C) | sum of squares(C)] of
    v = y (+ residual).  Attached to y as `_vnet_stats`; the batch-norm that normalises (y + residual) runs only its finalize."""
    __slots__ = ("partial", "rows", "residual")

    def __init__(self, partial, rows, residual):
        self.partial, self.rows, self.residual = partial, rows, residual


def _epilogue_stats_buffer(bf16, ks, kx, stride, x0, x1, O, dims_out):
    if _SYNC_BN is not None:                  # cross-replica statistics need the raw moments of the whole tensor: generic path
        return None
    L = _lib.lib()
    B, C0, C1 = x0.shape[0], x0.shape[-1], (x1.shape[-1] if x1 is not None else 0)
    if _is16(x0) and not bf16:                # bf16-storage 2^3 stride-2 convolution: one row per workgroup of the direct kernel,
        if O % 4:                             # or the generic fp32 MFMA kernel's brick rows
            return None
        rows = L.vnet_conv2_direct_stats_rows(C0, O, B, *dims_out) if _DIRECT2["on"] else 0
        if rows <= 0:
            rows = L.vnet_conv_stats_rows(ks, kx, stride, 0, C0 + C1, O, 0, B, *dims_out)
    elif bf16:
        # (the kernels that stage bf16 sources have their own brick shapes: one partial row per brick)
        rows = L.vnet_conv_b16_stats_rows(C0, C1, O, 0, B, *dims_out)     # (the deep-level kernel has its own bricks: csrc/conv_deep.h)
    else:
        # fp32 MFMA kernels: measured (profiles/r02_epilogue_stats.txt) the STATS instantiations lose in their main loop most of
        # what the statistics pass costs (+1..3 % per launch, residual re-read on the input conv): the fused form is worth
        # 0.07 ms of a 25.7 ms step; `_FUSE["bn_stats_fp32_direct"] = False` keeps it to the split-K launches (statistics from the reduce kernel)
        rows = 0
        # (NOT conditional on bn_stats_fp32_direct -- _ConvFn.forward takes the f32x3 kernel whenever _x3_ok, and the buffer
        #  must have that kernel's row count whatever the fp32-MFMA switch says)
        if ks == 5 and stride == 1 and kx in (0, 5) and _x3_ok(C0, C1, O, 0, B, dims_out):
            rows = L.vnet_conv_x3_stats_rows(C0 + C1, O, B, *dims_out)        # f32x3 kernel: one row per 2x8x16 brick (or per reduce block)
        if ks == 2 and stride == 2 and x1 is None and _DIRECT2["on"]:
            rows = L.vnet_conv2_direct_stats_rows(C0, O, B, *dims_out)        # the LDS-free direct kernel (levels 1-2): one row per workgroup
        if rows <= 0:
            if not _FUSE["bn_stats_fp32_direct"] and not L.vnet_conv_stats_from_reduce(ks, kx, stride, C0 + C1, O, B, *dims_out):
                return None
            rows = L.vnet_conv_stats_rows(ks, kx, stride, 0, C0 + C1, O, 0, B, *dims_out)
    if rows <= 0:
        return None
    return torch.empty((rows, 2 * O), dtype=torch.float32, device=x0.device)


def conv(x0, w, b, ks, stride=1, x1=None, bn_stats=False, bn_residual=None):
    """tf.nn.convolution(concat(x0,x1), w, 'SAME', strides) + b (reference layers2.py:63).
    bn_stats: the output feeds tf.layers.batch_normalization(y [+ bn_residual]) -- the convolution's epilogue also produces
    that batch-norm's partial sums (include/vnet_hip.h: vnet_conv_fwd_stats), so its statistics pass over y is not needed."""
    if x0.dim() != 5:
        raise NotImplementedError("only the 3-D (NDHWC) path is built; 2-D is out of scope (SURVEY section 2 row 11)")
    if _meta(x0):
        B, D, H, W, _ = x0.shape
        return torch.empty((B, _same_out(D, stride), _same_out(H, stride), _same_out(W, stride), w.shape[-1]), device="meta")
    _need_gpu(x0, "conv", allow16=True)
    stats = None
    if bn_stats and _FUSE["bn_stats"]:
        dims_out = tuple(_same_out(int(v), stride) for v in x0.shape[1:4])
        bf16 = ks == 5 and stride == 1 and _is16(x0)
        stats = _epilogue_stats_buffer(bf16, ks, 0, stride, x0, x1, w.shape[-1], dims_out)
    if stats is None:
        return _ConvFn.apply(x0, x1, w, b, ks, stride, False, None)
    y = _ConvFn.apply(x0, x1, w, b, ks, stride, False, None, stats, bn_residual)
    y._vnet_stats = _EpilogueStats(stats, stats.shape[0], bn_residual)
    return y


def conv_transpose2(x, w, b, out_spatial):
    """tf.nn.conv3d_transpose(x, w, output_shape, [1,2,2,2,1], 'SAME') + b (reference layers2.py:73)."""
    if x.dim() != 5:
        raise NotImplementedError("only the 3-D (NDHWC) path is built")
    if _meta(x):
        return torch.empty((x.shape[0],) + tuple(out_spatial) + (w.shape[-2],), device="meta")
    _need_gpu(x, "conv_transpose2", allow16=True)
    return _ConvFn.apply(x, None, w, b, 2, 2, True, tuple(out_spatial))


# ---- batch-norm (+residual, +tile, +activation) -------------------------------------------------------
# ---- cross-replica batch-norm ---------------------------------------------------------
# Default data-parallel semantics are per-replica statistics (== the reference run with BatchSize=1 per patch).
# With a group set here every batch-norm reduces its moments over all ranks, which reproduces the reference's
# single-device BatchSize=N numbers (networks.py:319 reduces over the batch axis as well).  Every rank must hold
# the same number of rows per layer (equal per-rank batch), as data-parallel training does.
_SYNC_BN = None      # (all_reduce callable, world size) or None


def set_sync_batch_norm(group=None, enabled=True):
    """Enable (or, with enabled=False, disable) cross-replica batch-norm statistics over `group`
    (default: the world group of torch.distributed)."""
    global _SYNC_BN
    if not enabled:
        _SYNC_BN = None
        return
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise VnetHipError("set_sync_batch_norm needs an initialised torch.distributed process group")
    world = dist.get_world_size(group)
    _SYNC_BN = ((lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)), world) if world > 1 else None


def _bn_statistics(L, x, r, bcast, M, C, mean, invstd, mm, mv, ws, nb, pre=None, r_orig=None):
    """mean/invstd (+ moving-average update) of s = x (+ r); returns the row count the statistics cover.
    pre: _EpilogueStats the producing convolution attached to x (used when it covers exactly x + r)."""
    if (_SYNC_BN is None and pre is not None and not bcast and pre.residual is r_orig and pre.partial.shape[1] == 2 * C):
        check(L.vnet_bn_finalize_partial(_ptr(pre.partial), pre.rows, C, float(M), BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(invstd),
                                         _ptr(mm), _ptr(mv), _stream()), "vnet_bn_finalize_partial")
        return float(M)
    x16 = _is16(x)
    if _SYNC_BN is None:
        if x16:
            check(L.vnet_bn_stats_b16(_ptr(x), _ptr(r), M, C, BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(invstd),
                                      _ptr(mm), _ptr(mv), _ptr(ws), nb, _stream()), "vnet_bn_stats_b16")
        else:
            check(L.vnet_bn_stats(_ptr(x), _ptr(r), int(bcast), M, C, BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(invstd),
                                  _ptr(mm), _ptr(mv), _ptr(ws), nb, _stream()), "vnet_bn_stats")
        return float(M)
    all_reduce, world = _SYNC_BN
    sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    if x16:
        check(L.vnet_bn_moments_b16(_ptr(x), _ptr(r), M, C, _ptr(sums), _ptr(ws), nb, _stream()), "vnet_bn_moments_b16")
    else:
        check(L.vnet_bn_moments(_ptr(x), _ptr(r), int(bcast), M, C, _ptr(sums), _ptr(ws), nb, _stream()), "vnet_bn_moments")
    all_reduce(sums)
    check(L.vnet_bn_finalize(_ptr(sums), float(M) * world, C, BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(invstd),
                             _ptr(mm), _ptr(mv), _stream()), "vnet_bn_finalize")
    return float(M) * world


# bf16 storage, tiny tensors (the 8^3 level: <= 512 rows): statistics + finalize + normalise in ONE launch, and reduce + finalize +
# apply in one (vnet_bn_small_*_b16; `_SMALL_BN["on"] = False`: the streaming kernels everywhere).  Measured (profiles/ab_env.sh, C5 step):
# <= 512 rows -0.015 ms, <= 1024 the same, <= 8192 (the 16^3 level too) +0.13 ms -- one workgroup per channel octet uses 16 bytes
# of every 256-byte row it touches, on 16-32 CUs; the five launch-bound streaming launches on 128+ workgroups are faster there.
_SMALL_BN = {"on": True, "rows": 512}


def _bn_small(M, C, *tensors):
    return (_SMALL_BN["on"] and M <= _SMALL_BN["rows"] and _SYNC_BN is None and all(t is None or _is16(t) for t in tensors)
            and bool(_lib.lib().vnet_bn_small_ok(int(M), int(C))))


class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, alpha, act, bcast, mm, mv):
        L = _lib.lib()
        ctx.slot_r = getattr(r, "_vnet_slot", None)
        pre, r_orig = getattr(x, "_vnet_stats", None), r
        x = x.contiguous()
        r = r.contiguous() if r is not None else None
        C = gamma.numel()
        M = x.numel() if bcast else x.numel() // C
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        ctx.small = small = (not bcast) and _is16(x) and _bn_small(M, C, x, r)
        if small:
            y = torch.empty(x.shape[:-1] + (C,), dtype=torch.bfloat16, device=dev)
            check(L.vnet_bn_small_fwd_b16(_ptr(x), _ptr(r), M, C, BN_EPS, BN_MOMENTUM, _ptr(gamma), _ptr(beta), act, _ptr(alpha),
                                          _ptr(mean), _ptr(invstd), _ptr(mm), _ptr(mv), _ptr(y), _stream()), "vnet_bn_small_fwd_b16")
            ctx.m_total, ctx.sync, ctx.b16 = float(M), None, True
            ctx.save_for_backward(x, r, gamma, beta, alpha, mean, invstd)
            ctx.params = (gamma, beta, alpha)
            ctx.cfg = (act, bcast, M, C)
            ctx.mark_non_differentiable(mean, invstd)
            ctx.set_materialize_grads(False)
            return y, mean, invstd
        nb = L.vnet_bn_ws_bytes(C)
        ws = workspace(nb, dev)
        ctx.m_total = _bn_statistics(L, x, r, bcast, M, C, mean, invstd, mm, mv, ws, nb, pre, r_orig)
        ctx.sync = _SYNC_BN
        # bf16-storage: a bf16 input, or the tiled fp32 1-channel image in that mode (C = 8 * 2^k; the 2..5-class batch-norm of the
        # logits stays on the fp32 kernels)
        ctx.b16 = b16 = _is16(x) or (bcast and _COMPUTE["store16"] and C % 8 == 0)
        if b16:
            if r is not None and not _is16(r):
                raise VnetHipError("bn_act: bf16 and float32 tensors mixed")
            y = torch.empty(x.shape[:-1] + (C,), dtype=torch.bfloat16, device=dev)
            check(L.vnet_bn_act_fwd_b16(_ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                                        act, _ptr(alpha), _ptr(y), _stream()), "vnet_bn_act_fwd_b16")
        else:
            y = torch.empty(x.shape[:-1] + (C,), dtype=torch.float32, device=dev)
            check(L.vnet_bn_act_fwd(_ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                                    act, _ptr(alpha), _ptr(y), _stream()), "vnet_bn_act_fwd")
        ctx.save_for_backward(x, r, gamma, beta, alpha, mean, invstd)
        ctx.params = (gamma, beta, alpha)
        ctx.cfg = (act, bcast, M, C)
        ctx.mark_non_differentiable(mean, invstd)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the (non-differentiable) statistics outputs
        return y, mean, invstd

    @staticmethod
    def backward(ctx, dy, _gmean, _ginvstd):
        L = _lib.lib()
        x, r, gamma, beta, alpha, mean, invstd = ctx.saved_tensors
        act, bcast, M, C = ctx.cfg
        dy = dy.contiguous()
        dev = dy.device
        gref, bref, aref = ctx.params
        dgamma, sg = _grad_out(gref)
        dbeta, sbt = _grad_out(bref)
        dalpha, sa = _grad_out(aref) if alpha is not None else (None, None)
        need_ds = ctx.needs_input_grad[0] or (r is not None and ctx.needs_input_grad[1])
        if ctx.b16:
            ds = torch.empty(dy.shape, dtype=torch.bfloat16, device=dev) if need_ds else None
        else:
            ds = torch.empty(dy.shape, dtype=torch.float32, device=dev) if need_ds else None
        nb = L.vnet_bn_ws_bytes(C)
        ws = workspace(nb, dev)
        if ctx.small:
            if not _is16(dy):
                raise VnetHipError("bn_act backward: expected a bfloat16 gradient")
            check(L.vnet_bn_small_bwd_b16(_ptr(dy), _ptr(x), _ptr(r), M, C, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), act,
                                          _ptr(alpha), _ptr(dgamma), _ptr(dbeta), _ptr(dalpha), _ptr(ds), _stream()), "vnet_bn_small_bwd_b16")
        elif ctx.b16:
            if not _is16(dy):
                raise VnetHipError("bn_act backward: expected a bfloat16 gradient")
            check(L.vnet_bn_act_bwd_reduce_b16(_ptr(dy), _ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd),
                                               _ptr(gamma), _ptr(beta), act, _ptr(alpha), _ptr(dgamma), _ptr(dbeta),
                                               _ptr(dalpha), _ptr(ws), nb, _stream()), "vnet_bn_act_bwd_reduce_b16")
            if ds is not None:
                if ctx.sync is None:
                    sdz, sdzx = dbeta, dgamma
                else:
                    tot = torch.cat([dbeta.reshape(-1), dgamma.reshape(-1)])
                    ctx.sync[0](tot)
                    sdz, sdzx = tot, tot[C:]
                check(L.vnet_bn_act_bwd_apply_b16(_ptr(dy), _ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd),
                                                  _ptr(gamma), _ptr(beta), act, _ptr(alpha), _ptr(sdz), _ptr(sdzx),
                                                  ctx.m_total, None, _ptr(ds), _stream()), "vnet_bn_act_bwd_apply_b16")
        elif ctx.sync is None:
            check(L.vnet_bn_act_bwd(_ptr(dy), _ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd), _ptr(gamma),
                                    _ptr(beta), act, _ptr(alpha), _ptr(dgamma), _ptr(dbeta), _ptr(dalpha), _ptr(ds),
                                    _ptr(ws), nb, _stream()), "vnet_bn_act_bwd")
        else:
            # this replica's parameter gradients stay local (the gradient all-reduce averages them);
            # the data gradient needs the sums over the whole cross-replica batch
            check(L.vnet_bn_act_bwd_reduce(_ptr(dy), _ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd),
                                           _ptr(gamma), _ptr(beta), act, _ptr(alpha), _ptr(dgamma), _ptr(dbeta),
                                           _ptr(dalpha), _ptr(ws), nb, _stream()), "vnet_bn_act_bwd_reduce")
            if ds is not None:
                tot = torch.cat([dbeta.reshape(-1), dgamma.reshape(-1)])
                ctx.sync[0](tot)
                check(L.vnet_bn_act_bwd_apply(_ptr(dy), _ptr(x), _ptr(r), int(bcast), M, C, _ptr(mean), _ptr(invstd),
                                              _ptr(gamma), _ptr(beta), act, _ptr(alpha), _ptr(tot), _ptr(tot[C:]),
                                              ctx.m_total, None, _ptr(ds), _stream()), "vnet_bn_act_bwd_apply")
        th = getattr(gref, "_vnet_deferred", None)
        if th is not None and sg is not None and sbt is not None:
            del gref._vnet_deferred
            th(1)                                          # the fused input conv adds its share of dgamma / dbeta (see _InputConvFn)
        dx = ds
        if bcast and ds is not None:
            dx = colsum_rows(ds)
        if r is not None and ds is not None and ctx.slot_r is not None and ctx.slot_r.first is None:
            ctx.slot_r.first = ds                      # the block input's other consumer (conv_1) adds its gradient into this
        return dx, (ds if r is not None else None), _grad_ret(dgamma, sg), _grad_ret(dbeta, sbt), _grad_ret(dalpha, sa), None, None, None, None


class _BnChainFn(torch.autograd.Function):
    """Decoder batch-norm chains in closed form (include/vnet_hip.h, vnet_bn_chain_*):
    kind 0: act(BN3(BN1(x) + BN2(BN1(x))))   (networks.py:333-337)      kind 1: act(BNb(x + BNa(x)))   (networks.py:358-361)
    One statistics pass + one normalise pass forward, one reduce + one apply pass backward."""

    @staticmethod
    def forward(ctx, x, kind, act, alpha, g1, b1, g2, b2, g3, b3, bufs):
        L = _lib.lib()
        pre = getattr(x, "_vnet_stats", None)
        x = x.contiguous()
        C = g1.numel()
        M = x.numel() // C
        dev = x.device
        mm1, mv1, mm2, mv2, mm3, mv3 = bufs
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        nb = L.vnet_bn_ws_bytes(C)
        ws = workspace(nb, dev)
        ctx.m_total = _bn_statistics(L, x, None, False, M, C, mean, invstd, mm1, mv1, ws, nb, pre, None)
        ctx.sync = _SYNC_BN
        ceff = torch.empty(C, dtype=torch.float32, device=dev)
        deff = torch.empty(C, dtype=torch.float32, device=dev)
        check(L.vnet_bn_chain_coef_fwd(kind, C, BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(invstd), _ptr(g1), _ptr(b1), _ptr(g2), _ptr(b2),
                                       _ptr(g3), _ptr(b3), _ptr(ceff), _ptr(deff), _ptr(mm2), _ptr(mv2), _ptr(mm3), _ptr(mv3),
                                       _stream()), "vnet_bn_chain_coef_fwd")
        ctx.b16 = _is16(x)
        if ctx.b16:
            y = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
            check(L.vnet_bn_act_fwd_b16(_ptr(x), None, 0, M, C, _ptr(mean), _ptr(invstd), _ptr(ceff), _ptr(deff),
                                        act, _ptr(alpha), _ptr(y), _stream()), "vnet_bn_act_fwd_b16")
        else:
            y = torch.empty(x.shape, dtype=torch.float32, device=dev)
            check(L.vnet_bn_act_fwd(_ptr(x), None, 0, M, C, _ptr(mean), _ptr(invstd), _ptr(ceff), _ptr(deff),
                                    act, _ptr(alpha), _ptr(y), _stream()), "vnet_bn_act_fwd")
        ctx.save_for_backward(x, alpha, g1, g2, g3, mean, invstd, ceff, deff)
        ctx.params = (alpha, g1, b1, g2, b2, g3, b3)
        ctx.cfg = (kind, act, M, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, alpha, g1, g2, g3, mean, invstd, ceff, deff = ctx.saved_tensors
        kind, act, M, C = ctx.cfg
        aref, g1r, b1r, g2r, b2r, g3r, b3r = ctx.params
        dy = dy.contiguous()
        dev = dy.device
        dC = torch.empty(C, dtype=torch.float32, device=dev)
        dD = torch.empty(C, dtype=torch.float32, device=dev)
        dalpha, sa = _grad_out(aref) if alpha is not None else (None, None)
        nb = L.vnet_bn_ws_bytes(C)
        ws = workspace(nb, dev)
        if ctx.b16:
            check(L.vnet_bn_act_bwd_reduce_b16(_ptr(dy), _ptr(x), None, 0, M, C, _ptr(mean), _ptr(invstd), _ptr(ceff), _ptr(deff),
                                               act, _ptr(alpha), _ptr(dC), _ptr(dD), _ptr(dalpha), _ptr(ws), nb, _stream()),
                  "vnet_bn_act_bwd_reduce_b16")
        else:
            check(L.vnet_bn_act_bwd_reduce(_ptr(dy), _ptr(x), None, 0, M, C, _ptr(mean), _ptr(invstd), _ptr(ceff), _ptr(deff),
                                           act, _ptr(alpha), _ptr(dC), _ptr(dD), _ptr(dalpha), _ptr(ws), nb, _stream()),
                  "vnet_bn_act_bwd_reduce")
        if ctx.sync is None:
            tot = None
            dCg, dDg = dC, dD
        else:                      # cross-replica statistics: the data gradient needs the sums over every replica
            tot = torch.cat([dD, dC])
            ctx.sync[0](tot)
            dDg, dCg = tot[:C], tot[C:]
        outs = [_grad_out(r) if r is not None else (None, None) for r in (g1r, b1r, g2r, b2r, g3r, b3r)]
        (dg1, s1), (db1, t1), (dg2, s2), (db2, t2), (dg3, s3), (db3, t3) = outs
        extra = torch.empty(C, dtype=torch.float32, device=dev)
        check(L.vnet_bn_chain_coef_bwd(kind, C, BN_EPS, ctx.m_total, _ptr(mean), _ptr(invstd), _ptr(g1), _ptr(g2), _ptr(g3),
                                       _ptr(dC), _ptr(dD), _ptr(dCg), _ptr(dg1), _ptr(db1), _ptr(dg2), _ptr(db2), _ptr(dg3), _ptr(db3),
                                       _ptr(extra), _stream()), "vnet_bn_chain_coef_bwd")
        dx = None
        if ctx.needs_input_grad[0] and ctx.b16:
            dx = torch.empty(dy.shape, dtype=torch.bfloat16, device=dev)
            check(L.vnet_bn_act_bwd_apply_b16(_ptr(dy), _ptr(x), None, 0, M, C, _ptr(mean), _ptr(invstd), _ptr(ceff), _ptr(deff),
                                              act, _ptr(alpha), _ptr(dDg), _ptr(dCg), ctx.m_total, _ptr(extra), _ptr(dx),
                                              _stream()), "vnet_bn_act_bwd_apply_b16")
        elif ctx.needs_input_grad[0]:
            dx = torch.empty(dy.shape, dtype=torch.float32, device=dev)
            check(L.vnet_bn_act_bwd_apply(_ptr(dy), _ptr(x), None, 0, M, C, _ptr(mean), _ptr(invstd), _ptr(ceff), _ptr(deff),
                                          act, _ptr(alpha), _ptr(dDg), _ptr(dCg), ctx.m_total, _ptr(extra), _ptr(dx),
                                          _stream()), "vnet_bn_act_bwd_apply")
        g3ret = _grad_ret(dg3, s3) if g3r is not None else None
        b3ret = _grad_ret(db3, t3) if b3r is not None else None
        return (dx, None, None, _grad_ret(dalpha, sa) if alpha is not None else None, _grad_ret(dg1, s1), _grad_ret(db1, t1),
                _grad_ret(dg2, s2), _grad_ret(db2, t2), g3ret, b3ret, None)


def bn_chain(x, kind, act, alpha, g1, b1, g2, b2, g3=None, b3=None, moving=(None,) * 6):
    """kind 0: act(BN3(BN1(x) + BN2(BN1(x)))) with (g1,b1),(g2,b2),(g3,b3); kind 1: act(BNb(x + BNa(x))) with (g1,b1)=a, (g2,b2)=b.
    moving = (mm1, mv1, mm2, mv2, mm3, mv3) moving-average buffers (None to skip)."""
    a = ACT[act]
    if _meta(x):
        return torch.empty(x.shape, device="meta")
    _need_gpu(x, "bn_chain", allow16=True)
    if a == 2 and alpha is None:
        raise VnetHipError("prelu needs alpha")
    if kind == 0 and (g3 is None or b3 is None):
        raise VnetHipError("bn_chain kind 0 needs three batch-norm layers")
    return _BnChainFn.apply(x, int(kind), a, alpha if a == 2 else None, g1, b1, g2, b2, g3, b3, tuple(moving))


def colsum_rows(ds):
    """Gradient of tf.tile over channels: sum over the channel axis (1-channel input)."""
    # rows are tiny (C floats); reuse the column-sum kernel on the transposed problem is not worth it:
    # the 1-channel image never requires grad in the reference (placeholder), so this is only reached in tests.
    return ds.sum(dim=-1, keepdim=True)


def bn_act(x, gamma, beta, act=None, alpha=None, residual=None, tile=False, moving_mean=None, moving_var=None, want_stats=False):
    """tf.layers.batch_normalization(x (+ residual), training=True) followed by `act`.
    tile=True: x has one channel and is broadcast to gamma.numel() channels (tf.tile, networks.py:258)."""
    a = ACT[act]
    if _meta(x):
        y = torch.empty(x.shape[:-1] + (gamma.numel(),), device="meta")
        return (y, None, None) if want_stats else y
    _need_gpu(x, "bn_act", allow16=True)
    if a == 2 and alpha is None:
        raise VnetHipError("prelu needs alpha")
    y, mean, invstd = _BnActFn.apply(x, residual, gamma, beta, alpha if a == 2 else None, a, bool(tile), moving_mean, moving_var)
    return (y, mean, invstd) if want_stats else y


def bn_update_only(x, C, moving_mean, moving_var):
    """A batch-norm layer whose output is unused ('dead', networks.py:358): only its moving-average
    update op runs (it is in UPDATE_OPS, model.py:665-666)."""
    if _meta(x):
        return
    L = _lib.lib()
    pre = getattr(x, "_vnet_stats", None)
    x = x.contiguous()
    M = x.numel() // C
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = L.vnet_bn_ws_bytes(C)
    ws = workspace(nb, x.device)
    _bn_statistics(L, x, None, False, M, C, mean, invstd, moving_mean, moving_var, ws, nb, pre, None)


# ---- stand-alone activation (API parity with layers2.prelu; the networks use the fused bn_act) ------------
class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha, act):
        L = _lib.lib()
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        check(L.vnet_act_fwd(_ptr(x), M, C, act, _ptr(alpha), _ptr(y), _stream()), "vnet_act_fwd")
        ctx.save_for_backward(x, alpha)
        ctx.act = act
        ctx.aref = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, alpha = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dalpha, sa = _grad_out(ctx.aref) if alpha is not None else (None, None)      # straight into the flat gradient buffer when there is one
        nb = L.vnet_bn_ws_bytes(C)
        ws = workspace(nb, x.device)
        check(L.vnet_act_bwd(_ptr(dy), _ptr(x), M, C, ctx.act, _ptr(alpha), _ptr(dalpha), _ptr(dx), _ptr(ws), nb, _stream()),
              "vnet_act_bwd")
        return dx, (_grad_ret(dalpha, sa) if alpha is not None else None), None


def activation(x, act, alpha=None):
    if _meta(x):
        return x
    _need_gpu(x, "activation")
    return _ActFn.apply(x, alpha if ACT[act] == 2 else None, ACT[act])


# ---- 1x1x1 head --------------------------------------------------------------------------------------------
class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        L = _lib.lib()
        x = x.contiguous()
        C, K = w.shape[-2], w.shape[-1]
        M = x.numel() // C
        y = torch.empty(x.shape[:-1] + (K,), dtype=torch.float32, device=x.device)        # logits stay fp32 in every mode
        if _is16(x):
            check(L.vnet_head_fwd_b16(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, C, K, _stream()), "vnet_head_fwd_b16")
        else:
            check(L.vnet_head_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, C, K, _stream()), "vnet_head_fwd")
        ctx.save_for_backward(x, w)
        ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        C, K = w.shape[-2], w.shape[-1]
        M = x.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, sw = _grad_out(ctx.params[0])
        db, sb = _grad_out(ctx.params[1])
        nb = L.vnet_head_ws_bytes(C, K)
        ws = workspace(nb, x.device)
        if _is16(x):
            check(L.vnet_head_bwd_b16(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), M, C, K, _ptr(ws), nb, _stream()),
                  "vnet_head_bwd_b16")
        else:
            check(L.vnet_head_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), M, C, K, _ptr(ws), nb, _stream()),
                  "vnet_head_bwd")
        return dx, _grad_ret(dw, sw), _grad_ret(db, sb)


def head_conv(x, w, b):
    """convolution(x, [1,1,1,C,K]) of the output layer (reference networks.py:298-302)."""
    if _meta(x):
        return torch.empty(x.shape[:-1] + (w.shape[-1],), device="meta")
    _need_gpu(x, "head_conv", allow16=True)
    return _HeadFn.apply(x, w, b)


# ---- fused softmax + Dice / cross-entropy loss ---------------------------------------------------------------
def parse_loss(name):
    """Loss.Name -> kind bits (reference model.py:495-558)."""
    valid = ("xent", "weighted_xent", "sorensen", "weighted_sorensen", "jaccard", "weighted_jaccard",
             "mixed_sorensen", "mixed_weighted_sorensen", "mixed_jaccard", "mixed_weighted_jaccard")
    if name not in valid:
        raise SystemExit("Invalid loss function")
    kind = LOSS_KIND["xent"] if name.endswith("xent") else (LOSS_KIND["sorensen"] if "sorensen" in name else LOSS_KIND["jaccard"])
    if "weighted" in name:
        kind |= LOSS_WEIGHTED
    if name.startswith("mixed_"):
        kind |= LOSS_MIXED
    return kind


_CONST_VEC = {}


def _const_vector(values, device):
    """Small constant vector (Loss.Weights) on the device, uploaded once: a pageable host-to-device copy is not allowed
    inside a stream capture."""
    key = (tuple(float(v) for v in values), device)
    t = _CONST_VEC.get(key)
    if t is None:
        t = _CONST_VEC[key] = torch.as_tensor(list(key[0]), dtype=torch.float32).to(device)
    return t


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, kind, weights, alpha, smooth, want_softmax, want_pred):
        L = _lib.lib()
        logits = logits.contiguous()
        labels = labels.contiguous()
        B, K = logits.shape[0], logits.shape[-1]
        V = logits.numel() // (B * K)
        dev = logits.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dice = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(2 * B * K + 1, dtype=torch.float32, device=dev)
        sm = torch.empty_like(logits) if want_softmax else None
        pred = torch.empty(logits.shape[:-1], dtype=torch.int64, device=dev) if want_pred else None
        nb = L.vnet_loss_ws_bytes(B, K)
        ws = workspace(nb, dev)
        check(L.vnet_softmax_dice_fwd(_ptr(logits), _ptr(labels), B, V, K, kind, _ptr(weights), alpha, smooth,
                                      _ptr(sm), _ptr(pred), _ptr(loss), _ptr(dice), _ptr(coef), _ptr(ws), nb, _stream()),
              "vnet_softmax_dice_fwd")
        ctx.save_for_backward(logits, labels, weights, coef)
        ctx.cfg = (kind, alpha, B, V, K)
        ctx.mark_non_differentiable(dice)
        ctx.set_materialize_grads(False)
        outs = [loss, dice]
        if sm is not None:
            ctx.mark_non_differentiable(sm)
        if pred is not None:
            ctx.mark_non_differentiable(pred)
        return loss, dice, sm, pred

    @staticmethod
    def backward(ctx, gloss, gdice, gsm, gpred):
        L = _lib.lib()
        logits, labels, weights, coef = ctx.saved_tensors
        kind, alpha, B, V, K = ctx.cfg
        g = gloss.contiguous().to(torch.float32)
        dl = torch.empty_like(logits)
        check(L.vnet_softmax_dice_bwd(_ptr(logits), _ptr(labels), B, V, K, kind, _ptr(weights), alpha, _ptr(coef),
                                      _ptr(g), _ptr(dl), _stream()), "vnet_softmax_dice_bwd")
        return dl, None, None, None, None, None, None, None


def softmax_loss(logits, labels, loss_name="sorensen", weights=None, alpha=1.0, smooth=1e-5,
                 want_softmax=False, want_pred=False):
    """softmax (model.py:447) + one_hot (model.py:474-477) + the loss switch (model.py:495-558).
    labels: int32 [B,D,H,W,1] or [B,D,H,W].  Returns (loss, dice_value, softmax|None, pred|None)."""
    _need_gpu(logits, "softmax_loss")
    kind = parse_loss(loss_name)
    K = logits.shape[-1]
    wt = None
    if kind & LOSS_WEIGHTED:
        if weights is None or len(weights) != K:
            raise AssertionError("Length of DICE weight is {}, should be {}".format(0 if weights is None else len(weights), K))
        wt = _const_vector(weights, logits.device)
    if labels.dtype != torch.int32:
        labels = labels.to(torch.int32)
    return _LossFn.apply(logits, labels, kind, wt, float(alpha), float(smooth), want_softmax, want_pred)


def softmax_argmax(logits):
    """Inference fetches of evaluate (model.py:914-917): 'softmax:0' and 'predicted_label/prediction:0'."""
    _need_gpu(logits, "softmax_argmax")
    lab = torch.zeros(logits.shape[:-1], dtype=torch.int32, device=logits.device)
    with torch.no_grad():
        _, _, sm, pred = _LossFn.apply(logits, lab, 0, None, 1.0, 1e-5, True, True)
    return sm, pred


# ---- dropout ------------------------------------------------------------------------------------------------------
class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed):
        L = _lib.lib()
        x = x.contiguous()
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        st = _STEP_STATE["active"]
        if _is16(x):
            y = torch.empty_like(x)
            check(L.vnet_dropout_fwd_b16(_ptr(x), _ptr(y), _ptr(mask), x.numel(), rate, seed, _ptr(st), _stream()), "vnet_dropout_fwd_b16")
        else:
            y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            check(L.vnet_dropout_fwd_dev(_ptr(x), _ptr(y), _ptr(mask), x.numel(), rate, seed, _ptr(st), _stream()), "vnet_dropout_fwd")
        ctx.save_for_backward(mask)
        ctx.rate = rate
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        if _is16(dy):
            check(L.vnet_dropout_bwd_b16(_ptr(dy), _ptr(mask), _ptr(dx), dy.numel(), ctx.rate, _stream()), "vnet_dropout_bwd_b16")
            return dx, None, None
        check(L.vnet_dropout_bwd(_ptr(dy), _ptr(mask), _ptr(dx), dy.numel(), ctx.rate, _stream()), "vnet_dropout_bwd")
        return dx, None, None


_DROP_SEED = [0x5EED, 0]


def dropout(x, rate):
    """tf.nn.dropout(x, rate=rate) (reference networks.py:321): identity when rate == 0."""
    rate = float(rate)
    if rate == 0.0 or _meta(x):
        return x
    _need_gpu(x, "dropout", allow16=True)
    if _STEP_STATE["active"] is not None:
        # graph-replayable: the seed of a layer is its position in the pass, the step number comes from the device state
        _DROP_SEED[1] += 1
        return _DropoutFn.apply(x, rate, 0x5EED0000 + _DROP_SEED[1])
    _DROP_SEED[0] += 1
    return _DropoutFn.apply(x, rate, _DROP_SEED[0])


def begin_dropout_pass():
    """Reset the per-pass layer counter of the state-driven dropout (called at the start of a step)."""
    _DROP_SEED[1] = 0


# ---- optimiser apply + sliding-window accumulate (no autograd) ----------------------------------------------------
def adam_apply(p, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, gscale=1.0, state=None):
    """state: device step-state buffer holding lr_t (then the `lr_t` argument is ignored) -- the graph-replayable form."""
    if state is not None:
        check(_lib.lib().vnet_adam_apply_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(state), beta1, beta2, eps, gscale,
                                             _stream()), "vnet_adam_apply_dev")
        return
    check(_lib.lib().vnet_adam_apply(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr_t, beta1, beta2, eps, gscale, _stream()),
          "vnet_adam_apply")


def sgd_apply(p, g, lr, gscale=1.0, state=None):
    if state is not None:
        check(_lib.lib().vnet_sgd_apply_dev(_ptr(p), _ptr(g), p.numel(), _ptr(state), gscale, _stream()), "vnet_sgd_apply_dev")
        return
    check(_lib.lib().vnet_sgd_apply(_ptr(p), _ptr(g), p.numel(), lr, gscale, _stream()), "vnet_sgd_apply")


def momentum_apply(p, g, acc, lr, momentum, nesterov=False, gscale=1.0, state=None):
    if state is not None:
        check(_lib.lib().vnet_momentum_apply_dev(_ptr(p), _ptr(g), _ptr(acc), p.numel(), _ptr(state), momentum, int(nesterov), gscale,
                                                 _stream()), "vnet_momentum_apply_dev")
        return
    check(_lib.lib().vnet_momentum_apply(_ptr(p), _ptr(g), _ptr(acc), p.numel(), lr, momentum, int(nesterov), gscale, _stream()),
          "vnet_momentum_apply")


def hard_metrics(pred, labels, K):
    """Accuracy and per-class tp/tn/fp/fn, sensitivity, specificity, hard Dice (reference model.py:588-626) from
    the K x K confusion matrix computed on the GPU.  pred int64 [...], labels int32 [...] (class indices)."""
    L = _lib.lib()
    pred = pred.contiguous().reshape(-1)
    labels = labels.to(torch.int32).contiguous().reshape(-1)
    cm = torch.empty(K * K, dtype=torch.float64, device=pred.device)
    nb = L.vnet_confusion_ws_bytes(K)
    ws = workspace(nb, pred.device)
    check(L.vnet_confusion_matrix(_ptr(pred), _ptr(labels), pred.numel(), K, _ptr(cm), _ptr(ws), nb, _stream()),
          "vnet_confusion_matrix")
    return metrics_from_confusion(cm.cpu().numpy().reshape(K, K))


# ---- tf.metrics.auc (reference model.py:607,613,624) ------------------------------------------------------------------
def tf_auc_thresholds(num_thresholds=200):
    """TF 1.15 metrics_impl.auc: kepsilon = 1e-7; [0 - eps] + [(i + 1) / (n - 1) for i in range(n - 2)] + [1 + eps], used as
    float32 constants against float32 predictions."""
    import numpy as np
    eps = 1e-7
    th = [0.0 - eps] + [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + eps]
    return np.asarray(th, dtype=np.float32)


def auc_histogram(softmax, labels, K, cls, num_thresholds=200):
    """Counts behind tf.metrics.auc(one_hot(labels)[..., cls], softmax[..., cls]): float64 [2][T+1] on the device --
    row 0 the voxels of class `cls`, row 1 all others; bin = number of thresholds strictly below the prediction."""
    L = _lib.lib()
    sm = softmax.contiguous()
    lab = labels.to(torch.int32).contiguous().reshape(-1)
    n = lab.numel()
    th = _const_vector(tf_auc_thresholds(num_thresholds), sm.device)
    hist = torch.empty((2, num_thresholds + 1), dtype=torch.float64, device=sm.device)
    nb = L.vnet_auc_ws_bytes(num_thresholds)
    ws = workspace(nb, sm.device)
    check(L.vnet_auc_histogram(_ptr(sm), _ptr(lab), n, K, cls, _ptr(th), num_thresholds, _ptr(hist), _ptr(ws), nb, _stream()),
          "vnet_auc_histogram")
    return hist


def auc_from_hist(hist):
    """ROC AUC the way tf.metrics.auc(curve='ROC', summation_method='trapezoidal') evaluates it from its per-threshold
    counters: tp[t] = #{positive, p > thr[t]}, rec = (tp + eps) / (tp + fn + eps), fpr = fp / (fp + tn + eps),
    auc = sum((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2)."""
    import numpy as np
    h = np.asarray(hist, dtype=np.float64)
    eps = 1e-7
    tp = h[0][::-1].cumsum()[::-1][1:]          # tp[t] = sum_{b > t} hist_pos[b]
    fp = h[1][::-1].cumsum()[::-1][1:]
    fn, tn = h[0].sum() - tp, h[1].sum() - fp
    rec = (tp + eps) / (tp + fn + eps)
    fpr = fp / (fp + tn + eps)
    return float(((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2.0).sum())


class StreamingMetrics(object):
    """The tf.metrics of reference model.py:588-626: accumulating (like TF's local variables, over every update since
    construction) confusion counts and AUC counters; result() gives accuracy of the LAST batch (model.py:590 is not
    streaming) and per class i > 0: tp/tn/fp/fn, sensitivity, specificity, hard Dice 2tp/(2tp+fp+fn) and the ROC AUC."""

    def __init__(self, K, num_thresholds=200):
        self.K, self.T = K, num_thresholds
        self.cm = None
        self.hist = [None] * K
        self.last_accuracy = None

    def update(self, pred, labels, softmax=None):
        import numpy as np
        m = hard_metrics(pred, labels, self.K)
        self.last_accuracy = m["accuracy"]
        self.cm = m["confusion"] if self.cm is None else self.cm + m["confusion"]
        if softmax is not None:
            for c in range(1, self.K):
                h = auc_histogram(softmax, labels, self.K, c, self.T).cpu().numpy()
                self.hist[c] = h if self.hist[c] is None else self.hist[c] + h
        return self

    def result(self):
        out = metrics_from_confusion(self.cm)
        out["accuracy"] = self.last_accuracy
        for c in range(1, self.K):
            if self.hist[c] is not None:
                out[c]["auc"] = auc_from_hist(self.hist[c])
        return out


def metrics_from_confusion(cm):
    K = cm.shape[0]
    n = cm.sum()
    out = {"accuracy": float(cm.trace() / max(n, 1.0)), "confusion": cm}
    for c in range(K):
        tp = cm[c, c]; fn = cm[c].sum() - tp; fp = cm[:, c].sum() - tp; tn = n - tp - fn - fp
        out[c] = {"tp": tp, "tn": tn, "fp": fp, "fn": fn,
                  "sensitivity": float(tp / max(tp + fn, 1e-30)), "specificity": float(tn / max(tn + fp, 1e-30)),
                  "dice": float(2 * tp / max(2 * tp + fp + fn, 1e-30))}
    return out


def accumulate_patch(patch, vol, count, origin):
    """vol[z0:z0+pz, ...] += patch ; count += 1 (reference model.py:919-929)."""
    pz, py, px, K = patch.shape
    D, H, W = vol.shape[:3]
    check(_lib.lib().vnet_accumulate_patch(_ptr(patch.contiguous()), _ptr(vol), _ptr(count), K, pz, py, px,
                                           int(origin[0]), int(origin[1]), int(origin[2]), D, H, W, _stream()),
          "vnet_accumulate_patch")

"""ctypes binding of libvnet_hip.so (include/vnet_hip.h).

The library is the product: there is NO fallback.  If the shared object is missing, or a
kernel reports an error, this module raises -- nothing silently routes to PyTorch or the CPU.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VNET_HIP_LIB") or os.path.join(_HERE, "libvnet_hip.so")      # (override: A/B of kernel builds)
CSRC = os.path.join(_HERE, "csrc")

_vp, _i, _i64, _f, _sz, _u64, _d = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                    ctypes.c_size_t, ctypes.c_uint64, ctypes.c_double)

# name -> (restype, argtypes) : must list every symbol include/vnet_hip.h declares
SIGNATURES = {
    "vnet_version": (ctypes.c_char_p, []),
    "vnet_set_option": (_d, [ctypes.c_char_p, _d]),
    "vnet_get_option": (_d, [ctypes.c_char_p]),
    "vnet_packed_weight_floats": (_sz, [_i, _i, _i, _i]),
    "vnet_pack_weights": (_i, [_i, _vp, _vp, _i, _i, _i, _vp]),
    "vnet_packed_dims": (_i, [_i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "vnet_pack_weights_batched": (_i, [_vp, _i, _vp]),
    "vnet_conv_x3_ok": (_i, [_i] * 8),
    "vnet_conv_x3_stats_rows": (_i, [_i] * 6),
    "vnet_conv_x3_ws_bytes": (_sz, [_i] * 6),
    "vnet_conv_fwd_x3": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_wgrad_x3_ok": (_i, [_i] * 7),
    "vnet_wgrad_x3_ws_bytes": (_sz, [_i] * 6),
    "vnet_conv_wgrad_x3": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_conv_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "vnet_conv_fwd": (_i, [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i,
                           _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_conv_fwd_acc": (_i, [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i,
                               _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_conv_stats_rows": (_i, [_i] * 11),
    "vnet_conv_stats_from_reduce": (_i, [_i] * 9),
    "vnet_conv_fwd_stats": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i,
                                 _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "vnet_bn_finalize_partial": (_i, [_vp, _i, _i, _d, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "vnet_colsum_b16_ws_bytes": (_sz, [_i]),
    "vnet_colsum_b16": (_i, [_vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "vnet_conv_b16_ws_bytes": (_sz, [_i] * 8),
    "vnet_conv_b16_stats_rows": (_i, [_i] * 8),
    "vnet_wgrad_bf16_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "vnet_wgrad_defer": (_i, [_i, _vp]),
    "vnet_wgrad_pending": (_i, [_vp]),
    "vnet_wgrad_flush": (_i, [_vp]),
    "vnet_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "vnet_conv_wgrad": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp,
                             _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_tile_im2col_x": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vnet_input_conv_fold": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "vnet_input_conv_grads": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vnet_input_conv_direct_ok": (_i, [_i, _i, _i, _i, _i]),
    "vnet_input_conv_direct_stats_rows": (_i, [_i, _i, _i, _i]),
    "vnet_input_conv_fold_border": (_i, [_vp, _i, _vp, _vp, _vp]),
    "vnet_input_conv_direct_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vnet_input_wgrad_direct_slabs": (_i, [_i, _i, _i, _i]),
    "vnet_input_wgrad_direct": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_head_fwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "vnet_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _sz, _vp]),
    "vnet_head_ws_bytes": (_sz, [_i, _i]),
    "vnet_colsum_ws_bytes": (_sz, [_i]),
    "vnet_colsum": (_i, [_vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "vnet_bn_ws_bytes": (_sz, [_i]),
    "vnet_bn_stats": (_i, [_vp, _vp, _i, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_bn_moments": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _sz, _vp]),
    "vnet_bn_finalize": (_i, [_vp, _d, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "vnet_bn_act_bwd_reduce": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_bn_act_bwd_apply": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp]),
    "vnet_bn_chain_coef_fwd": (_i, [_i, _i, _f, _f] + [_vp] * 14 + [_vp]),
    "vnet_bn_chain_coef_bwd": (_i, [_i, _i, _f, _d] + [_vp] * 15 + [_vp]),
    "vnet_bn_act_fwd": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "vnet_bn_act_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp,
                             _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_act_fwd": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp]),
    "vnet_act_bwd": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_loss_ws_bytes": (_sz, [_i, _i]),
    "vnet_softmax_dice_fwd": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_softmax_dice_bwd": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _f, _vp, _vp, _vp, _vp]),
    "vnet_dice_coe_fwd": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _f, _vp, _vp, _vp, _sz, _vp]),
    "vnet_dice_coe_bwd": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "vnet_dropout_fwd": (_i, [_vp, _vp, _vp, _i64, _f, _u64, _vp]),
    "vnet_dropout_bwd": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    "vnet_adam_apply": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp]),
    "vnet_sgd_apply": (_i, [_vp, _vp, _i64, _f, _f, _vp]),
    "vnet_momentum_apply": (_i, [_vp, _vp, _vp, _i64, _f, _f, _i, _f, _vp]),
    "vnet_step_state_set": (_i, [_vp, _f, _f, _u64, _vp]),
    "vnet_adam_apply_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp]),
    "vnet_sgd_apply_dev": (_i, [_vp, _vp, _i64, _vp, _f, _vp]),
    "vnet_momentum_apply_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _f, _i, _f, _vp]),
    "vnet_dropout_fwd_dev": (_i, [_vp, _vp, _vp, _i64, _f, _u64, _vp, _vp]),
    "vnet_confusion_ws_bytes": (_sz, [_i]),
    "vnet_confusion_matrix": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _sz, _vp]),
    "vnet_auc_ws_bytes": (_sz, [_i]),
    "vnet_auc_histogram": (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "vnet_accumulate_patch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    # bf16-storage mode
    "vnet_cast_bf16": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "vnet_conv_fwd_b16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_conv_fwd_b16_padded": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "vnet_conv_wgrad_b16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_wgrad_job_bytes": (_sz, []),
    "vnet_conv_wgrad_b16_group": (_i, [_vp, _i, _vp]),
    "vnet_conv2_fwd_b16": (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "vnet_conv2_wgrad_b16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vnet_conv2_direct_ok": (_i, [_i, _i]),
    "vnet_conv2_direct_stats_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "vnet_conv2_direct_b16": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vnet_conv2_direct_f32": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vnet_bn_stats_b16": (_i, [_vp, _vp, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_bn_moments_b16": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _sz, _vp]),
    "vnet_bn_act_fwd_b16": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "vnet_bn_act_bwd_reduce_b16": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vnet_bn_act_bwd_apply_b16": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp]),
    "vnet_bn_small_ok": (_i, [_i64, _i]),
    "vnet_bn_small_fwd_b16": (_i, [_vp, _vp, _i64, _i, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vnet_bn_small_bwd_b16": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vnet_head_fwd_b16": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "vnet_head_bwd_b16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _sz, _vp]),
    "vnet_dropout_fwd_b16": (_i, [_vp, _vp, _vp, _i64, _f, _u64, _vp, _vp]),
    "vnet_dropout_bwd_b16": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
}

class WgradJob(ctypes.Structure):
    """include/vnet_hip.h: vnet_wgrad_job (one layer of vnet_conv_wgrad_b16_group)."""
    _fields_ = [("x0", ctypes.c_void_p), ("x1", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p),
                ("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_size_t),
                ("C0", ctypes.c_int), ("C1", ctypes.c_int), ("Cout", ctypes.c_int), ("Cin_dw", ctypes.c_int),
                ("B", ctypes.c_int), ("D", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("ks", ctypes.c_int)]


ERRORS = {-1: "VNET_E_BADARG", -2: "VNET_E_UNSUPPORTED", -3: "VNET_E_WORKSPACE"}


def set_option(name, value):
    """vnet_set_option: returns the previous value.  name without the VNET_ prefix, e.g. set_option("BF16_DEEP", 0)."""
    prev = lib().vnet_set_option(name.encode(), float(value))
    if prev != prev:
        raise VnetHipError("unknown library option %r" % (name,))
    for c in _MEMOS:                      # the planners read the options (F32_SMALL, BF16_DEEP, X3_*): memoised size queries are stale
        c.clear()
    return prev


class VnetHipError(RuntimeError):
    pass


def build(force=False):
    """Compile libvnet_hip.so in-tree with hipcc for gfx950 (works without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    res = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise VnetHipError("building libvnet_hip.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VnetHipError("libvnet_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc, gfx950).  The HIP library is the only compute path." % LIB_PATH)
        # torch ships its own libamdhip64: import it FIRST so the kernels, torch's streams and its
        # allocator live in ONE HIP runtime (loading ours first would bind /opt/rocm's copy instead).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
            if name == "vnet_conv_b16_stats_rows":
                setattr(L, name, _memo(fn, (b"BF16_DEEP", b"BF16_DEEP_TARGET"), L))     # (the kernel choice follows these options)
            elif name.endswith("_ws_bytes") or name.endswith("_stats_rows") or name == "vnet_conv_stats_from_reduce" or name == "vnet_packed_weight_floats":
                setattr(L, name, _memo(fn))    # pure size queries, asked before every launch: answer repeats from a dict
        _lib = L
    return _lib


_MEMOS = []


def _memo(fn, opts=None, L=None):
    cache = {}
    _MEMOS.append(cache)

    def cached(*args):
        key = args if opts is None else args + tuple(L.vnet_get_option(o) for o in opts)
        r = cache.get(key)
        if r is None:
            r = cache[key] = fn(*args)
        return r
    return cached


def check(code, what):
    if code != 0:
        raise VnetHipError("%s failed: %s" % (what, ERRORS.get(code, "hipError_t %d" % code)))

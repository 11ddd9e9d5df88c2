"""fp32 5^3 convolutions on the small volumes of the deep levels: us per launch incl. the split-K reduce, for the
brick choices of option F32_SMALL (0: 8x8x8 bricks / 8 waves, 1: 4x8x8 / 4 waves, 2: + 4x4x4 bricks for volumes <= 4^3).
python profiles/bench_small.py [iters]"""
import sys
import torch
sys.path.insert(0, ".")
from vnet_tensorflow_amd import ops, _lib

dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SHAPES = [(2, 4, 256, 256), (2, 8, 128, 128), (1, 8, 256, 256), (2, 8, 256, 128), (2, 8, 128, 256), (1, 8, 128, 128), (2, 4, 128, 128), (1, 4, 256, 256)]
ops.set_compute_dtype("fp32")
for B, P, ci, co in SHAPES:
    x = torch.randn(B, P, P, P, ci, device=dev)
    w = torch.randn(5, 5, 5, ci, co, device=dev) * 0.05
    y = torch.empty(B, P, P, P, co, device=dev)
    wp = ops.packed_weights(w, ops.PACK_FWD, 125, ci, co)
    fl = 2.0 * B * P ** 3 * 125 * ci * co
    res, ref = [], None
    for mode in (0, 1, 2):
        _lib.set_option("F32_SMALL", mode)
        f = lambda: ops._conv_call(5, 1, 0, x, None, wp, None, y, None, (P, P, P), (P, P, P))
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        if ref is None:
            ref = y.clone()
        res.append((us, fl / us / 1e6, float((y - ref).abs().max() / ref.abs().max())))
    print("conv k5 %d^3x%d %3d->%3d  " % (P, B, ci, co) + "   ".join("F32_SMALL=%d %6.1f us %6.1f TF/s (d %.1e)" % ((k,) + r) for k, r in zip((0, 1, 2), res)), flush=True)
_lib.set_option("F32_SMALL", 2)

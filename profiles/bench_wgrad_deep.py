"""Deep-level 5^3 filter gradients in bf16 storage, one layer per launch (incl. its slab reduce): the round-3 kernels against the
z-streaming kernel (csrc/wgrad_zs.h, VNET_WGRAD_ZS=1), then the grouped launch of the C5 step's 17 deep layers:
   python profiles/bench_wgrad_deep.py [iters]"""
import os
import sys
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops, _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda', 0)
SHAPES = [(64, 32, 0, 32), (64, 32, 32, 32), (32, 64, 0, 64), (32, 64, 64, 64), (16, 128, 0, 128), (16, 128, 128, 128), (8, 256, 0, 256)]


def timeit(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def tensors(P, c0, c1, co):
    x0 = torch.randn(1, P, P, P, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(1, P, P, P, c1, device=dev).to(torch.bfloat16) if c1 else None
    dy = torch.randn(1, P, P, P, co, device=dev).to(torch.bfloat16)
    dw = torch.empty(5, 5, 5, c0 + c1, co, device=dev)
    return x0, x1, dy, dw


for (P, c0, c1, co) in SHAPES:
    x0, x1, dy, dw = tensors(P, c0, c1, co)
    out = []
    for zs in ("0", "1"):
        _lib.set_option("WGRAD_ZS", int(zs))
        out.append(timeit(lambda: ops._wgrad5_b16_call(x0, x1, dy, dw, (P, P, P), c0 + c1)))
    _lib.set_option("WGRAD_ZS", 2)
    fl = 2.0 * P ** 3 * 125 * (c0 + c1) * co
    print("wgrad-b16 %2d^3 %3d->%3d   round-3 kernel %6.1f us %7.1f TF/s   z-streaming %6.1f us %7.1f TF/s   x%.2f" % (
        P, c0 + c1, co, out[0] * 1e3, fl / out[0] / 1e9, out[1] * 1e3, fl / out[1] / 1e9, out[0] / out[1]), flush=True)

# the deep layers of the C5 step (levels 3-5 of encoder and decoder + bottom): one grouped launch + one batched reduce
LAYERS = [(32, 64, 0, 64)] * 5 + [(32, 64, 64, 64)] + [(16, 128, 0, 128)] * 5 + [(16, 128, 128, 128)] + [(8, 256, 0, 256)] * 3
ts = [tensors(*s) for s in LAYERS]
sinks = [ops.GradSink(t[3]) for t in ts]
fl = sum(2.0 * P ** 3 * 125 * (c0 + c1) * co for (P, c0, c1, co) in LAYERS)


def group():
    with ops.deferred_wgrad_reduce():
        for (P, c0, c1, co), (x0, x1, dy, dw), s in zip(LAYERS, ts, sinks):
            ops._wgrad5_b16_call(x0, x1, dy, dw, (P, P, P), c0 + c1, owner=s)


for label, env in (("every layer on its own (round 3)", {"VNET_WGRAD_GROUP": "0"}),
                   ("grouped, round-3 kernel bodies", {"VNET_WGRAD_ZS": "0"}), ("grouped, z-streaming everywhere", {"VNET_WGRAD_ZS": "1"}), ("grouped, z-streaming below 32^3 (default)", {})):
    _lib.set_option("WGRAD_ZS", int(env.get("VNET_WGRAD_ZS", 2)))
    ops.set_wgrad_group(env.get("VNET_WGRAD_GROUP", "1") != "0")
    t = timeit(group)
    print("15 deep layers of C5, %-42s %7.1f us %7.1f TF/s" % (label, t * 1e3, fl / t / 1e9), flush=True)

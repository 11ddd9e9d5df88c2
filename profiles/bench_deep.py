"""Deep-level 5^3 convolutions in bf16 storage, deep kernel (csrc/conv_deep.h) against the generic kernels, one process:
   python profiles/bench_deep.py [iters]
Prints us per convolution INCLUDING its split-K reduce launch (what a layer costs in the step) and TF/s."""
import os
import sys
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops, _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda', 0)
SHAPES = [(32, 64, 0, 64), (32, 64, 64, 64), (32, 64, 0, 128), (16, 128, 0, 128), (16, 128, 128, 128), (16, 128, 0, 256), (8, 256, 0, 256)]
if os.environ.get("BENCH_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["BENCH_SHAPES"].split(";")]


def bench(P, c0, c1, co, deep):
    _lib.set_option("BF16_DEEP", int(deep))          # (round 5: the library reads its switches once; flips go through vnet_set_option)
    x0 = torch.randn(1, P, P, P, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(1, P, P, P, c1, device=dev).to(torch.bfloat16) if c1 else None
    w = torch.randn(5, 5, 5, c0 + c1, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev)
    wp = ops.packed_weights(w, ops.PACK_FWD_BF16, 125, c0 + c1, co)
    y = torch.empty(1, P, P, P, co, device=dev, dtype=torch.bfloat16)

    def run():
        ops._conv5_b16_call(x0, x1, wp, b, y, None, (P, P, P))
    for _ in range(5):
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
    return best, y


for (P, c0, c1, co) in SHAPES:
    t0, y0 = bench(P, c0, c1, co, "0")
    t1, y1 = bench(P, c0, c1, co, "1")
    fl = 2.0 * P ** 3 * 125 * (c0 + c1) * co
    print("conv-b16 %2d^3 %3d->%3d   generic %6.1f us %7.1f TF/s   deep %6.1f us %7.1f TF/s   x%.2f" % (
        P, c0 + c1, co, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, t0 / t1), flush=True)

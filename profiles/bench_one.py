"""Micro-benchmark of ONE 5^3 convolution problem in one mode (for rocprofv3 --pmc passes):
   python profiles/bench_one.py <conv|wgrad> <fp32|fp32_split3|bf16> P Cin Cout [iters]
   bf16 = bf16 storage (bf16 tensors in and out)"""
import sys
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops

kind, mode, P, ci, co = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device('cuda', 0)
ops.set_compute_dtype(mode)
import os
x = torch.randn(1, P, P, P, ci, device=dev)
w = torch.nn.Parameter(torch.randn(5, 5, 5, ci, co, device=dev) * 0.05)
b = torch.zeros(co, device=dev)
dy = torch.randn(1, P, P, P, co, device=dev)
dw = torch.empty(5, 5, 5, ci, co, device=dev)
if os.environ.get("BENCH_ZERO"):          # DVFS probe: all-zero operands toggle no datapath bits (MI355X_MICROARCH.md "DVFS give-back")
    x.zero_(); dy.zero_()
    with torch.no_grad():
        w.zero_()
if mode == 'bf16':
    x, dy = x.to(torch.bfloat16), dy.to(torch.bfloat16)


def run():
    if kind == 'conv':
        with torch.no_grad():
            ops._ConvFn.apply(x, None, w, b, 5, 1, False, None)
    elif mode == 'bf16':
        ops._wgrad5_b16_call(x, None, dy, dw, (P, P, P), ci)
    elif mode == 'fp32_split3':
        ops._wgrad_x3_call(x, None, dy, dw, (P, P, P))
    else:
        ops._wgrad_call(5, 1, x, None, dy, dw, (P, P, P), (P, P, P))


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("%s %s %d^3 %d->%d  %.3f ms  %.1f TF/s" % (kind, mode, P, ci, co, ms, 2.0 * P ** 3 * 125 * ci * co / ms / 1e9))

"""The single-modality input block alone: forward (+ statistics with residual) and filter gradient at 128^3, 16 channels.
python profiles/bench_input_block.py [reps]      (VNET_INPUT_DIRECT=0: rounds 1-5's x-im2col + 5x5x1 MFMA kernels; VNET_HIP_LIB: another build)"""
import sys
import torch
sys.path.insert(0, ".")
from vnet_tensorflow_amd import ops

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
P, C = 128, 16
img = torch.randn(1, P, P, P, 1, device=dev) * 40 + 120
g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
w = (torch.randn(5, 5, 5, C, C, device=dev) * 0.1).requires_grad_(True)
bi = torch.randn(C, device=dev).requires_grad_(True)
x, mean, invstd = ops.bn_act(img, g, b, None, None, None, True, None, None, want_stats=True)
dy = torch.randn(1, P, P, P, C, device=dev)


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fwd():
    with torch.no_grad():
        return ops.input_conv(img, g, b, mean, invstd, w, bi, bn_stats=True, bn_residual=x)


def both():
    y = ops.input_conv(img, g, b, mean, invstd, w, bi, bn_stats=True, bn_residual=x)
    y.backward(dy)
    w.grad = None; bi.grad = None


tf = timed(fwd)
tb = timed(both)
print("input block 128^3 1->%d: forward (fold + conv + statistics) %.3f ms, forward + backward %.3f ms (backward alone %.3f)" % (C, tf, tb, tb - tf))

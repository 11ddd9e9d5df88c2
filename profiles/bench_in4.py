"""The C5 input conv (4 modalities zero-padded to 8 channels -> 16) at 128^3, bf16 storage: x-im2col form of the 16-cout kernel
(vnet_conv_fwd_b16_padded) against the plain one, and the filter gradient.   python profiles/bench_in4.py"""
import sys
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops

dev = torch.device('cuda', 0)
ops.set_compute_dtype('bf16')
P = 128
x = ops.cast_input(torch.randn(1, P, P, P, 4, device=dev))
w = torch.nn.Parameter(torch.randn(5, 5, 5, 4, 16, device=dev) * 0.05)
b = torch.zeros(16, device=dev)
dy = torch.randn(1, P, P, P, 16, device=dev).to(torch.bfloat16)
dw = torch.empty(5, 5, 5, 4, 16, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(2):
    for on in (False, True):
        ops._IN4["on"] = on
        with torch.no_grad():
            ms = timed(lambda: ops._ConvFn.apply(x, None, w, b, 5, 1, False, None))
        print("forward  x-im2col=%-5s %.3f ms" % (on, ms))
    ms = timed(lambda: ops._wgrad5_b16_call(x, None, dy, dw, (P, P, P), 4))
    print("filter gradient          %.3f ms" % ms)

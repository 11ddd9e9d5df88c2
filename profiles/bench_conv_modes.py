import sys, torch, numpy as np
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops
dev = torch.device('cuda', 0)
cases = [(128,16,32),(128,32,16),(64,32,32),(64,64,32),(32,64,64),(32,128,64),(16,128,128),(16,256,128),(8,256,256)]
for mode in ('bf16', 'fp32'):
    ops.set_compute_dtype(mode)
    for P, ci, co in cases:
        x = torch.randn(1, P, P, P, ci, device=dev); w = (torch.randn(5,5,5,ci,co, device=dev)*0.05).requires_grad_(False)
        b = torch.zeros(co, device=dev)
        wparam = torch.nn.Parameter(w)
        with torch.no_grad():
            for _ in range(3): y = ops._ConvFn.apply(x, None, wparam, b, 5, 1, False, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): y = ops._ConvFn.apply(x, None, wparam, b, 5, 1, False, None)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * P**3 * 125 * ci * co
        dy = torch.randn_like(y); dw = torch.empty_like(w)
        for _ in range(3):
            (ops._wgrad_bf16_call(x, None, dy, dw, (P, P, P)) if mode == 'bf16' else ops._wgrad_call(5, 1, x, None, dy, dw, (P, P, P), (P, P, P)))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            (ops._wgrad_bf16_call(x, None, dy, dw, (P, P, P)) if mode == 'bf16' else ops._wgrad_call(5, 1, x, None, dy, dw, (P, P, P), (P, P, P)))
        e1.record(); torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 10
        print("%s %3d^3 %3d->%3d  conv %.3f ms %.1f TF/s | wgrad %.3f ms %.1f TF/s" % (mode, P, ci, co, ms, fl / ms / 1e9, ms2, fl / ms2 / 1e9), flush=True)

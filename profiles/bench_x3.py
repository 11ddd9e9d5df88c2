"""Micro-benchmark of the f32x3 convolution kernel against the fp32-MFMA kernel on the shapes VERDICT r4 #1 names.
python profiles/bench_x3.py [reps] [P,ci,co ...]      (e.g. 50 8,256,256 16,128,128: the deep levels)"""
import sys
import torch
sys.path.insert(0, ".")
from vnet_tensorflow_amd import ops

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(128, 32, 16), (128, 16, 32), (128, 16, 16), (64, 32, 32), (64, 64, 32), (64, 32, 64), (32, 64, 64), (32, 128, 64), (32, 64, 128)]
for P, ci, co in shapes:
    x = torch.randn(1, P, P, P, ci, device=dev)
    w = (torch.randn(5, 5, 5, ci, co, device=dev) * 0.05).requires_grad_(False)
    y = torch.empty(1, P, P, P, co, device=dev)
    flops = 2.0 * P ** 3 * 125 * ci * co
    out = []
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        if mode == "fp32":
            wp = ops.packed_weights(w, ops.PACK_FWD, 125, ci, co)
            f = lambda: ops._conv_call(5, 1, 0, x, None, wp, None, y, None, (P, P, P), (P, P, P))
        else:
            wp = ops.packed_weights(w, ops.PACK_FWD_X3, 125, ci, co)
            f = lambda: ops._conv_x3_call(x, None, wp, None, y, None, (P, P, P))
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append((mode, ms, flops / ms / 1e9))
        if mode == "fp32":
            yref = y.clone()
    # filter gradient of the same layer
    dy = torch.randn(1, P, P, P, co, device=dev)
    dw = torch.empty(5, 5, 5, ci, co, device=dev)
    outw = []
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        f = (lambda: ops._wgrad_call(5, 1, x, None, dy, dw, (P, P, P), (P, P, P))) if mode == "fp32" else (lambda: ops._wgrad_x3_call(x, None, dy, dw, (P, P, P)))
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        outw.append((ms, flops / ms / 1e9))
        if mode == "fp32":
            dwref = dw.clone()
    errw = float((dw - dwref).norm() / dwref.norm())
    err = float((y - yref).norm() / yref.norm())
    print("%3d^3 %3d->%3d  " % (P, ci, co) + "  ".join("%s %.3f ms %.1f TF/s" % o for o in out) + "  x%.2f  rel(x3 - fp32) %.2e" % (out[0][1] / out[1][1], err)
          + " | wgrad %.3f ms %.1f TF/s -> %.3f ms %.1f TF/s x%.2f rel %.2e" % (outw[0] + outw[1] + (outw[0][0] / outw[1][0], errw)), flush=True)
ops.set_compute_dtype("fp32")

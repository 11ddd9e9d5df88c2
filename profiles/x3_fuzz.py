"""Random-shape cross-check of the f32x3 kernels (both brick geometries) against the fp32-MFMA kernels on the GPU: forward with bias,
backward-data and filter gradient of one 5^3 layer per draw, rel-L2 <= 3e-6 (both sides are fp32-accurate: ~1e-6 each against fp64).
python profiles/x3_fuzz.py [draws] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from vnet_tensorflow_amd import ops

dev = torch.device("cuda", 0)
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = 0.0
for n in range(draws):
    W = int(rng.choice([8, 8, 8, 16, 20, 32, 7]))
    B, D, H = int(rng.integers(1, 4)), int(rng.integers(1, 21)), int(rng.integers(1, 21))
    C0 = 16 * int(rng.integers(1, 5)); C1 = 16 * int(rng.integers(0, 3)); Co = 16 * int(rng.integers(1, 5))
    if n % 7 == 0:
        C0, C1, Co = 128, 0, 64                          # enough chunks for the K split
    out = {}
    torch.manual_seed(n)
    x0 = torch.randn(B, D, H, W, C0, device=dev); x1 = torch.randn(B, D, H, W, C1, device=dev) if C1 else None
    w = torch.randn(5, 5, 5, C0 + C1, Co, device=dev) * 0.1; b = torch.randn(Co, device=dev); dy = torch.randn(B, D, H, W, Co, device=dev)
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        ops._X3["force"] = mode == "fp32_split3"
        a0 = x0.clone().requires_grad_(True); a1 = x1.clone().requires_grad_(True) if C1 else None
        ww = w.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
        ops.profile_start()
        y = ops.conv(a0, ww, bb, 5, 1, x1=a1)
        y.backward(dy)
        recs = [r[0] for r in ops.profile_stop()]
        if mode == "fp32_split3":
            assert sum(r.startswith("conv-x3") for r in recs) == 2 and sum(r.startswith("wgrad-x3") for r in recs) == 1, recs
        out[mode] = [y.detach(), a0.grad, ww.grad] + ([a1.grad] if C1 else [])
    ops._X3["force"] = False
    ops.set_compute_dtype("fp32")
    errs = [float((p - q).norm() / q.norm()) for p, q in zip(out["fp32_split3"], out["fp32"])]
    worst = max(worst, max(errs))
    flag = "" if max(errs) < 3e-6 else "   <-- FAIL"
    print("B%d %2dx%2dx%2d %3d+%2d->%3d  y %.2e dx %.2e dw %.2e%s" % (B, D, H, W, C0, C1, Co, errs[0], errs[1], errs[2], flag))
    assert max(errs) < 3e-6
print("draws %d, worst rel-L2(f32x3 - fp32 MFMA) %.2e" % (draws, worst))

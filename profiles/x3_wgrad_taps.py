"""Per-tap error of the f32x3 filter gradient against the fp64 oracle (benign operands, 32 x 64 x 128, 16 -> 16): is the excess over the
fp32 MFMA kernel uniform over the taps or tied to the kernel's tap -> wave / window-slot assignment?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vnet_oracle as O  # noqa: E402
from tests.util import g, rel_l2  # noqa: E402
from vnet_tensorflow_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
D, H, W, C, Co = 32, 64, 128, 16, 16
rng = np.random.default_rng(3)
f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
x, dy = f(rng.standard_normal((1, D, H, W, C))), f(rng.standard_normal((1, D, H, W, Co)))
w = f(rng.standard_normal((5, 5, 5, C, Co)) * 0.1)
_, dw_ref = O.conv_nd_bwd(x, w, dy, 1, need_dx=False)
res = {}
for mode in ("fp32", "fp32_split3"):
    ops.set_compute_dtype(mode)
    ops._X3["force"] = mode == "fp32_split3"
    tx, tw = g(x, dev), g(w, dev).requires_grad_(True)
    y = ops.conv(tx, tw, None, 5, 1)
    y.backward(g(dy, dev))
    torch.cuda.synchronize()
    res[mode] = tw.grad.cpu().numpy().astype(np.float64)
    ops._X3["force"] = False
    ops.set_compute_dtype("fp32")
scale = np.sqrt((dw_ref ** 2).mean())
for mode in res:
    e = (res[mode] - dw_ref) / scale                      # error in units of the rms filter-gradient element
    print(mode, "rel-L2 %.3e  mean signed error %.3e (units of rms |dw|)  rms %.3e" % (rel_l2(res[mode], dw_ref), e.mean(), np.sqrt((e ** 2).mean())))
    pt = np.sqrt((e ** 2).mean(axis=(3, 4)))            # [dz][dy][dx]
    print("  per-tap rms error: min %.2e median %.2e max %.2e" % (pt.min(), np.median(pt), pt.max()))
    for dz in range(5):
        print("   dz=%d " % dz + " | ".join(" ".join("%.1e" % pt[dz, dyy, dx] for dx in range(5)) for dyy in range(5)))
    print("  mean signed error per tap (dz=2 plane):", " ".join("%+.1e" % v for v in e.mean(axis=(3, 4))[2].ravel()))
d = (res["fp32_split3"] - res["fp32"]) / scale
print("f32x3 - fp32: rms %.3e mean %.3e" % (np.sqrt((d ** 2).mean()), d.mean()))

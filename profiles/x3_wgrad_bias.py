"""Why is the f32x3 FILTER GRADIENT less accurate than the fp32 MFMA's on operands with per-element exponent spread, and only there?
(profiles/r06_x3_adversarial_chain.txt: ratio 1.3 -> 3.4 as the accumulation chain grows 512 -> 4096 voxels, while the convolutions sit
at 0.59.)  Same shape, four operand classes, both kernels: rel-L2, and the SIGNED error along the reference (a bias shows as a
non-zero mean of (got - ref) in units of rms |ref|).  Second line per class: the forward convolution of the same x."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vnet_oracle as O  # noqa: E402
from tests.util import g, rel_l2  # noqa: E402
from vnet_tensorflow_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
D, H, W, C, Co = 32, 64, 128, int(os.environ.get("X3_C", 16)), int(os.environ.get("X3_C", 16))
f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
for kind in ("benign", "x spread", "dy spread", "both spread", "both spread 2^+-8", "both spread, per row"):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, D, H, W, C)); dy = rng.standard_normal((1, D, H, W, Co))
    S = 8 if "2^+-8" in kind else 20
    if kind.startswith(("x", "both")):
        x = x * np.exp2(rng.integers(-S, S + 1, (1, D, H, W, 1) if "row" not in kind else (1, D, H, 1, 1)))
    if kind.startswith(("dy", "both")):
        dy = dy * np.exp2(rng.integers(-S, S + 1, (1, D, H, W, 1) if "row" not in kind else (1, D, H, 1, 1)))
    x, dy = f(x), f(dy)
    w = f(rng.standard_normal((5, 5, 5, C, Co)) * 0.1)
    _, dw_ref = O.conv_nd_bwd(x, w, dy, 1, need_dx=False)
    y_ref = O.conv_nd_fwd(x, w, 1)
    out, outy = [], []
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        ops._X3["force"] = mode == "fp32_split3"
        tx, tw = g(x, dev), g(w, dev).requires_grad_(True)
        y = ops.conv(tx, tw, None, 5, 1)
        y.backward(g(dy, dev))
        torch.cuda.synchronize()
        got = tw.grad.cpu().numpy().astype(np.float64)
        e = (got - dw_ref)
        out.append((rel_l2(got, dw_ref), float(e.mean() / np.sqrt((dw_ref ** 2).mean())), float(np.median(np.abs(e) / np.abs(dw_ref)))))
        ey = y.detach().cpu().numpy().astype(np.float64) - y_ref
        outy.append((rel_l2(y.detach().cpu().numpy(), y_ref), float(ey.mean() / np.sqrt((y_ref ** 2).mean()))))
        ops._X3["force"] = False
        ops.set_compute_dtype("fp32")
    print("%-24s dw: fp32 MFMA rel-L2 %.2e mean err/rms %+.2e median %.2e | f32x3 rel-L2 %.2e mean err/rms %+.2e median %.2e | ratio %.2f" %
          ((kind,) + out[0] + out[1] + (out[1][0] / out[0][0],)), flush=True)
    print("%-24s  y: fp32 MFMA rel-L2 %.2e mean err/rms %+.2e | f32x3 rel-L2 %.2e mean err/rms %+.2e | ratio %.2f" %
          ((kind,) + outy[0] + outy[1] + (outy[1][0] / outy[0][0],)), flush=True)

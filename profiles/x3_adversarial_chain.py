"""f32x3 against the fp32 MFMA on the `spread` operand class (tests/test_hip_x3.py: every element of x, w, dy with its own exponent in
2^+-20) as a function of the ACCUMULATION CHAIN LENGTH -- the number of products one accumulator register takes in a row before partial
sums meet.  Convolution: chain = 125 x Cin (both kernels, when neither splits K); filter gradient: chain = voxels per workgroup.
Small volumes make the fp32-MFMA planner split K (plan_conv: nsplit x nz partial slabs) and give short filter-gradient chains; the
bench sizes have chains of 2000-4000 (convolution) and 16384 (128^3 filter gradient).  rel-L2 against the numpy-fp64 oracle.
    python profiles/x3_adversarial_chain.py > gpurun_out/r06_x3_adversarial_chain.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vnet_oracle as O  # noqa: E402
from tests.util import g, rel_l2  # noqa: E402
from vnet_tensorflow_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
print("%-22s %-9s %10s %10s | %-28s | %-28s | %s" % ("shape", "C->Co", "conv slabs", "wgrad chain", "fp32 MFMA fwd / dx / dw", "f32x3 fwd / dx / dw", "ratio fwd / dx / dw"))
for (D, H, W, C, Co) in ((6, 16, 32, 32, 32), (16, 32, 64, 32, 32), (32, 32, 64, 32, 32), (32, 64, 128, 16, 16), (64, 128, 128, 16, 16)):
    rng = np.random.default_rng(3)
    f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    x = f(rng.standard_normal((1, D, H, W, C)) * np.exp2(rng.integers(-20, 21, (1, D, H, W, C))))
    w = f(rng.standard_normal((5, 5, 5, C, Co)) * np.exp2(rng.integers(-20, 21, (5, 5, 5, C, Co))))
    y_ref = O.conv_nd_fwd(x, w, 1)
    dy = f(rng.standard_normal(y_ref.shape) * np.exp2(rng.integers(-20, 21, y_ref.shape)))
    dx_ref, dw_ref = O.conv_nd_bwd(x, w, dy, 1)
    slabs = L.vnet_conv_ws_bytes(5, 0, 1, 0, C, Co, 1, D, H, W) // (D * H * W * Co * 4)
    nbrick = -(-D // 2) * -(-H // 8) * -(-W // 16)
    nsplit = max(1, min(nbrick, -(-256 // ((C // 16) * (Co // 16)))))
    errs = {}
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        ops._X3["force"] = mode == "fp32_split3"
        tx, tw = g(x, dev).requires_grad_(True), g(w, dev).requires_grad_(True)
        y = ops.conv(tx, tw, None, 5, 1)
        y.backward(g(dy, dev))
        torch.cuda.synchronize()
        errs[mode] = (rel_l2(y.detach().cpu().numpy(), y_ref), rel_l2(tx.grad.cpu().numpy(), dx_ref), rel_l2(tw.grad.cpu().numpy(), dw_ref))
        ops._X3["force"] = False
        ops.set_compute_dtype("fp32")
    a, b = errs["fp32"], errs["fp32_split3"]
    print("%-22s %-9s %10d %10d | %.2e %.2e %.2e   | %.2e %.2e %.2e   | %.2f %.2f %.2f" %
          ("%dx%dx%d" % (D, H, W), "%d->%d" % (C, Co), max(1, slabs), nbrick // nsplit * 256, a[0], a[1], a[2], b[0], b[1], b[2], b[0] / a[0], b[1] / a[1], b[2] / a[2]), flush=True)

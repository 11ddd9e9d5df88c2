"""-m gpu: the f32x3 convolution kernels (csrc/conv_x3.h; ComputeDtype "fp32_split3") -- fp32 tensors, every product formed from six
bf16 products of exactly split operands -- against the numpy-fp64 oracle at the SAME 2e-6 the fp32-MFMA kernels are held to
(tests/test_hip_ops.py::test_conv5).  Reference call sites: layers2.py:59-63 (networks.py:316,333,346), autodiff model.py:660."""
import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.util import g, check_close, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture
def split3():
    from vnet_tensorflow_amd import ops
    ops.set_compute_dtype("fp32_split3")
    ops._X3["force"] = True
    yield ops
    ops._X3["force"] = False
    ops.set_compute_dtype("fp32")


@pytest.mark.parametrize("shape", [
    (1, 4, 8, 16, 16, 0, 16),      # whole bricks, one chunk, one cout block
    (2, 5, 9, 17, 16, 16, 16),     # ragged dims, two-source (concat), batch 2, two chunks
    (1, 8, 8, 16, 32, 0, 32),      # two cout blocks
    (1, 6, 16, 32, 64, 0, 16),     # four chunks: every rotation of the tap columns
    (1, 3, 5, 7, 16, 0, 48),       # a volume smaller than one brick, three cout blocks
    (1, 2, 8, 40, 48, 0, 32),      # three chunks, ragged x
    (1, 8, 8, 16, 128, 0, 32),     # few items, eight chunks: the chunks split over four workgroups per item (partial slabs + reduce)
    (1, 4, 16, 16, 32, 32, 32),    # K split with two sources
])
def test_conv5_x3(dev, split3, shape):
    ops = split3
    B, D, H, W, C0, C1, Co = shape
    rng = np.random.default_rng(sum(shape) + 11)
    x0 = rng.standard_normal((B, D, H, W, C0))
    x1 = rng.standard_normal((B, D, H, W, C1)) if C1 else None
    w = rng.standard_normal((5, 5, 5, C0 + C1, Co)) * 0.1
    b = rng.standard_normal(Co)
    xcat = x0 if x1 is None else np.concatenate((x0, x1), -1)
    y_ref = O.conv_nd_fwd(xcat, w, 1) + b
    dy = rng.standard_normal(y_ref.shape)
    dx_ref, dw_ref = O.conv_nd_bwd(xcat, w, dy, 1)
    tx0 = g(x0, dev).requires_grad_(True)
    tx1 = g(x1, dev).requires_grad_(True) if C1 else None
    tw, tb = g(w, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
    ops.profile_start()
    y = ops.conv(tx0, tw, tb, 5, 1, x1=tx1)
    tag = "conv-x3 [%d,%d,%d,%d] %d+%d->%d" % shape
    check_close(tag + " fwd", y, y_ref, 2e-6)
    y.backward(g(dy, dev))
    recs = ops.profile_stop()
    assert sum(1 for r in recs if r[0].startswith("conv-x3")) == 2, [r[0] for r in recs]     # forward and backward-data took the f32x3 kernel
    assert sum(1 for r in recs if r[0].startswith("wgrad-x3")) == 1, [r[0] for r in recs]    # and so did the filter gradient
    check_close(tag + " dx0", tx0.grad, dx_ref[..., :C0], 2e-6)
    if C1:
        check_close(tag + " dx1", tx1.grad, dx_ref[..., C0:], 2e-6)
    check_close(tag + " dw", tw.grad, dw_ref, 2e-6)


def test_x3_wide_dynamic_range(dev, split3):
    """The split is exact whatever the magnitudes: operands spread over 2^+-20 -- the result keeps fp32 accuracy (a bf16 or a
    two-piece product would be off by 2^-9 / 2^-17)."""
    ops = split3
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 4, 8, 16, 16)) * np.exp2(rng.integers(-20, 21, (1, 4, 8, 16, 16)))
    w = rng.standard_normal((5, 5, 5, 16, 16)) * np.exp2(rng.integers(-20, 21, (5, 5, 5, 16, 16)))
    xf, wf = x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64)
    y_ref = O.conv_nd_fwd(xf, wf, 1)
    y = ops.conv(g(xf, dev), g(wf, dev), None, 5, 1)
    assert rel_l2(y.cpu().numpy(), y_ref) < 2e-6


def test_x3_epilogue_statistics_accumulate_and_residual(dev, split3):
    """The batch-norm partial sums of the epilogue (rows per 2x8x16 brick), the residual in front of them, and y += conv."""
    ops = split3
    from vnet_tensorflow_amd import _lib
    L = _lib.lib()
    B, D, H, W, C, Co = 1, 5, 9, 20, 32, 32
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, D, H, W, C)); w = rng.standard_normal((5, 5, 5, C, Co)) * 0.1
    b = rng.standard_normal(Co); r = rng.standard_normal((B, D, H, W, Co)); y_old = rng.standard_normal((B, D, H, W, Co))
    y_ref = O.conv_nd_fwd(x, w, 1) + b
    tx, tw, tb, tr = g(x, dev), g(w, dev), g(b, dev), g(r, dev)
    wp = ops.packed_weights(tw, ops.PACK_FWD_X3, 125, C, Co)
    rows = L.vnet_conv_x3_stats_rows(C, Co, B, D, H, W)
    assert rows == 3 * 2 * 2
    stats = torch.full((rows, 2 * Co), float("nan"), device=dev)
    y = torch.empty((B, D, H, W, Co), device=dev)
    ops._conv_x3_call(tx, None, wp, tb, y, None, (D, H, W), stats=stats, res=tr)
    check_close("x3 stats fwd", y, y_ref, 2e-6)
    v = (y_ref + r).reshape(-1, Co)
    s = stats.cpu().numpy().astype(np.float64)
    assert np.isfinite(s).all()
    np.testing.assert_allclose(s[:, :Co].sum(0), v.sum(0), rtol=0, atol=2e-5 * np.abs(v).sum(0).max())
    np.testing.assert_allclose(s[:, Co:].sum(0), (v * v).sum(0), rtol=2e-6)
    ty = g(y_old, dev)
    ops._conv_x3_call(tx, None, wp, tb, ty, None, (D, H, W), accum=True)
    check_close("x3 accumulate", ty, y_ref + y_old, 2e-6)


def test_x3_split_is_exact(dev, split3):
    """h + m + l == w EXACTLY (8 + 8 + 8 significant bits; the remainders are formed by v_dot2c_f32_bf16): read the three pieces back
    from the packed filter image [k chunk][pair 65][n block][piece 3][64 lanes][8 k] and add them in float64."""
    ops = split3
    rng = np.random.default_rng(12)
    I = O = 16
    w = (rng.standard_normal((5, 5, 5, I, O)) * np.exp2(rng.integers(-30, 31, (5, 5, 5, I, O)))).astype(np.float32)
    w.reshape(-1)[:7] = [0.0, -0.0, 1.0, -1.0, 3.0e38, 1.0e-30, 2.0 ** -100]      # (not within 2^24 of the smallest normal: the last piece would be denormal)
    tw = g(w, dev)
    img = ops.packed_weights(tw, ops.PACK_FWD_X3, 125, I, O).cpu().numpy().view(np.uint16).reshape(65, 3, 64, 8)
    f = (img.astype(np.uint32) << 16).view(np.float32).astype(np.float64)        # bf16 -> float
    total = f.sum(1)                                                             # [pair][lane][8 k]
    w3 = w.reshape(125, I, O).astype(np.float64)
    for p in range(65):
        for hi in range(2):
            if p < 50:
                zp, r = divmod(p, 25); dx, dy = divmod(r, 5); dz = 2 * zp + hi; valid = True
            else:
                dx, q = divmod(p - 50, 3); dz = 4; dy = 2 * q + hi; valid = not (q == 2 and hi)
            tap = (dz * 5 + dy) * 5 + dx if valid else None
            for half in range(2):
                lanes = np.arange(16) + 16 * (half + 2 * hi)
                want = w3[tap, half * 8:half * 8 + 8, :].T if valid else np.zeros((16, 8))      # [n][k]
                got = total[p, lanes, :]
                bad = np.argwhere(got != want)
                assert bad.size == 0, (p, hi, half, [(tuple(b), got[tuple(b)], want[tuple(b)]) for b in bad[:4]])


def test_backward_runs_in_the_context_of_the_forward(dev):
    """a backward pass driven from OUTSIDE the model's OpsContext (a user calling loss.backward() after a forward that
    ran inside it) must take the forward's compute mode, pack registry and streams -- here: a fp32_split3 context, backward called
    while the default (fp32) context is current; bit-identical to the backward called inside the context, and not the fp32 kernels'
    bits."""
    from vnet_tensorflow_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 32, 32, 64, 16, generator=gen).to(dev)          # (256 bricks of 2x8x16: vnet_conv_x3_ok)
    dy = torch.randn(1, 32, 32, 64, 16, generator=gen).to(dev)
    w0 = (torch.randn(5, 5, 5, 16, 16, generator=gen) * 0.1).to(dev)
    b0 = torch.randn(16, generator=gen).to(dev)
    ctx = ops.OpsContext()
    with ops.context(ctx):
        ops.set_compute_dtype("fp32_split3")
    grads = {}
    for where in ("outside", "inside", "fp32"):
        xx, w, b = x.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        if where == "fp32":
            y = ops.conv(xx, w, b, 5, 1)
            y.backward(dy)
        else:
            with ops.context(ctx):
                y = ops.conv(xx, w, b, 5, 1)
                if where == "inside":
                    y.backward(dy)
            if where == "outside":
                assert ops.get_compute_dtype() == "fp32"
                y.backward(dy)
        torch.cuda.synchronize()
        grads[where] = (y.detach().clone(), xx.grad.clone(), w.grad.clone())
    for a, b_ in zip(grads["outside"], grads["inside"]):
        assert torch.equal(a, b_)
    assert not torch.equal(grads["outside"][1], grads["fp32"][1])          # (the f32x3 kernels really ran: other bits than the fp32 MFMA)
    assert rel_l2(grads["outside"][2].cpu().numpy(), grads["fp32"][2].cpu().numpy()) < 2e-6

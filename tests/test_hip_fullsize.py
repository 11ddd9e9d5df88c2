"""-m gpu: the hot path at BASELINE's FULL sizes (configs C3/C4: 128^3 patch, 16 base channels; C5 arithmetic).

The numpy oracle cannot evaluate a 128^3 network in test time, so parity at full size is established through
(a) the oracle on CROPS of the full-size tensors (a stride-1 SAME convolution restricted to a block depends only on
    the block + 2 voxels of halo, and a filter gradient with dy supported on a block only on that neighbourhood),
(b) exact known answers (delta filter, all-ones filter: integers, bit-exact in fp32),
(c) size-independent identities of the domain: linearity, adjointness <conv(x),y> = <x,conv^T(y)>, the filter
    gradient as the adjoint in w, batch-norm output moments, softmax/Dice closed forms, batch-duplication invariance
    of train-mode batch-norm through the WHOLE network, run-to-run determinism.
Tolerances are fp32-roundoff class and written at each check."""
import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.util import g, check_close, rel_l2

pytestmark = pytest.mark.gpu
P = 128


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _scale(a, b):
    """Natural size of <a,b> for random-sign data: the inner products below cancel to ~1 % of this, so the fp32
    roundoff of the two sides is measured against it, not against the (accidentally small) value itself."""
    return float(a.double().norm() * b.double().norm())


def _crop_blocks():
    # (z0, y0, x0) of 12^3 output blocks: a corner (SAME padding on three faces), an edge, the interior, the far corner
    return [(0, 0, 0), (0, 58, 116), (57, 61, 50), (116, 116, 116)]


def _oracle_block(xcat, w, z0, y0, x0, rb=None):
    """Oracle conv output on the 12^3 block at (z0,y0,x0) of a [1,P,P,P,C] tensor (numpy), via a crop with halo."""
    lo = [max(0, c - 2) for c in (z0, y0, x0)]
    hi = [min(P, c + 14) for c in (z0, y0, x0)]
    crop = xcat[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2], :].astype(np.float64)
    if rb is not None:
        crop, w = rb(crop), rb(w)
    y = O.conv_nd_fwd(crop, w.astype(np.float64), 1)
    # positions whose 5^3 window lies inside the crop or outside the VOLUME (true zero padding) are valid
    s = [c - l for c, l in zip((z0, y0, x0), lo)]
    return y[:, s[0]:s[0] + 12, s[1]:s[1] + 12, s[2]:s[2] + 12, :]


@pytest.mark.parametrize("mode", ["fp32", "fp32_split3"])
@pytest.mark.parametrize("C0,C1,Co", [(16, 0, 16), (16, 16, 16), (16, 0, 32)])
def test_conv5_full_resolution_against_oracle_crops(dev, mode, C0, C1, Co):
    """The north-star kernel (5^3 conv, 16/32 channels @128^3): forward, backward-data and filter gradient against the
    oracle on crops, rel-L2 2e-6 -- the fp32 MFMA kernels and (round 5) the f32x3 kernels: exactly split bf16 operands, six
    products each, the SAME bound.  (bf16 storage at this size: tests/test_hip_golden_full.py, teacher-forced layers.)"""
    from vnet_tensorflow_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(1234 + C0 + C1 + Co)
    x0 = torch.randn(1, P, P, P, C0, generator=gen)
    x1 = torch.randn(1, P, P, P, C1, generator=gen) if C1 else None
    w = torch.randn(5, 5, 5, C0 + C1, Co, generator=gen) * 0.05
    b = torch.randn(Co, generator=gen)
    rb = None
    tx0 = x0.to(dev).requires_grad_(True)
    tx1 = x1.to(dev).requires_grad_(True) if C1 else None
    tw, tb = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    ops.set_compute_dtype(mode)
    try:
        y = ops.conv(tx0, tw, tb, 5, 1, x1=tx1)
        xcat = x0.numpy() if x1 is None else np.concatenate((x0.numpy(), x1.numpy()), -1)
        for (z0, y0, xx0) in _crop_blocks():
            ref = _oracle_block(xcat, w.numpy(), z0, y0, xx0, rb) + b.numpy().astype(np.float64)
            got = y[:, z0:z0 + 12, y0:y0 + 12, xx0:xx0 + 12, :]
            check_close("conv %s block %s" % (mode, (z0, y0, xx0)), got, ref, 2e-6)
        # backward with dy supported on one 10^3 block: dx is supported on the block + halo, dw sees only x around it
        bz, by, bx = 60, 3, 115
        dy = torch.zeros_like(y)
        dyb = torch.randn(1, 10, 10, 10, Co, generator=gen)
        dy[:, bz:bz + 10, by:by + 10, bx:bx + 10, :] = dyb.to(dev)
        y.backward(dy)
    finally:
        ops.set_compute_dtype("fp32")
    lo = [max(0, c - 2) for c in (bz, by, bx)]
    hi = [min(P, c + 12) for c in (bz, by, bx)]
    xc = xcat[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2], :].astype(np.float64)
    dyc = np.zeros(xc.shape[:-1] + (Co,))
    s = [c - l for c, l in zip((bz, by, bx), lo)]
    dyc[:, s[0]:s[0] + 10, s[1]:s[1] + 10, s[2]:s[2] + 10, :] = dyb.numpy()
    wn = w.numpy().astype(np.float64)
    if rb is None:
        dx_ref, dw_ref = O.conv_nd_bwd(xc, wn, dyc, 1)
    else:
        dx_ref, _ = O.conv_nd_bwd(xc, rb(wn), rb(dyc), 1)
        _, dw_ref = O.conv_nd_bwd(rb(xc), wn, rb(dyc), 1)
    # the crop is zero-padded by the oracle where the volume continues: that is exact here because dy is zero there
    dxg = torch.cat((tx0.grad, tx1.grad), -1) if C1 else tx0.grad
    check_close("dx block", dxg[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2], :], dx_ref, 2e-6)
    outside = dxg.clone()
    outside[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2], :] = 0
    assert float(outside.abs().max()) == 0.0                      # nothing leaks outside the receptive field
    check_close("dw", tw.grad, dw_ref, 2e-6)
    check_close("db", tb.grad, dyb.numpy().reshape(-1, Co).sum(0), 2e-6)


def test_conv5_full_resolution_exact_known_answers(dev):
    """Integer-valued cases are exact in fp32, so these are BIT-exact checks of the index math at 128^3."""
    from vnet_tensorflow_amd import ops
    C = 16
    x = torch.randn(1, P, P, P, C, device=dev)
    w = torch.zeros(5, 5, 5, C, C, device=dev)
    w[2, 2, 2] = torch.eye(C, device=dev)
    b = torch.arange(C, device=dev, dtype=torch.float32)
    y = ops.conv(x, w, b, 5, 1)
    assert torch.equal(y, x + b)                                   # delta filter returns the input + bias
    ones = torch.ones(1, P, P, P, C, device=dev)
    y = ops.conv(ones, torch.ones(5, 5, 5, C, C, device=dev), torch.zeros(C, device=dev), 5, 1)
    cnt1 = torch.tensor([3, 4] + [5] * (P - 4) + [4, 3], device=dev, dtype=torch.float32)     # taps inside per axis
    ref = (cnt1[:, None, None] * cnt1[None, :, None] * cnt1[None, None, :] * C)[None, ..., None].expand_as(y)
    assert torch.equal(y, ref)                                     # 27*C at corners ... 125*C inside
    # down conv on even dims touches every input voxel exactly once; up conv of one voxel = the filter block
    xd = torch.ones(1, P, P, P, C, device=dev)
    wd = torch.ones(2, 2, 2, C, 2 * C, device=dev)
    yd = ops.conv(xd, wd, torch.zeros(2 * C, device=dev), 2, 2)
    assert yd.shape == (1, P // 2, P // 2, P // 2, 2 * C) and torch.equal(yd, torch.full_like(yd, 8.0 * C))
    xu = torch.zeros(1, P // 2, P // 2, P // 2, 2 * C, device=dev)
    xu[0, 5, 6, 7, 3] = 1.0
    wu = torch.randn(2, 2, 2, C, 2 * C, device=dev)
    yu = ops.conv_transpose2(xu, wu, torch.zeros(C, device=dev), (P, P, P))
    blk = yu[0, 10:12, 12:14, 14:16, :]
    assert torch.equal(blk, wu[:, :, :, :, 3])
    yu[0, 10:12, 12:14, 14:16, :] = 0
    assert float(yu.abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["fp32", "fp32_split3"])
def test_conv_identities_full_resolution(dev, mode):
    """Linearity in x, adjointness of backward-data, and the filter gradient as the adjoint in w, with fp64 dot
    products over all 128^3 x 16..32 values (2e-6 of |a||b|: sums of 3e7 fp32 terms), fp32 MFMA and f32x3 kernels."""
    from vnet_tensorflow_amd import ops
    Ci, Co = 32, 16
    q = lambda t: t
    x, z = q(torch.randn(1, P, P, P, Ci, device=dev)), q(torch.randn(1, P, P, P, Ci, device=dev))
    w = q(torch.randn(5, 5, 5, Ci, Co, device=dev) * 0.05).requires_grad_(True)
    dw_dir = q(torch.randn(5, 5, 5, Ci, Co, device=dev) * 0.05)
    zero_b = torch.zeros(Co, device=dev)
    yv = q(torch.randn(1, P, P, P, Co, device=dev))
    ops.set_compute_dtype(mode)
    try:
        xr = x.clone().requires_grad_(True)
        cx = ops.conv(xr, w, zero_b, 5, 1)
        cz = ops.conv(z, w, zero_b, 5, 1)
        if True:
            lin = ops.conv(2.0 * x - 0.5 * z, w, zero_b, 5, 1)
            assert rel_l2(lin.detach().cpu().numpy(), (2.0 * cx - 0.5 * cz).detach().cpu().numpy()) < 1e-5
        cx.backward(yv)
        lhs = _dot(cx.detach(), yv)
        assert abs(lhs - _dot(x, xr.grad)) <= 2e-6 * _scale(cx.detach(), yv)    # <conv(x), y> = <x, conv^T(y)>
        cdir = ops.conv(x, dw_dir, zero_b, 5, 1)                                 # conv is linear in w as well
        lhs = _dot(cdir.detach(), yv)
        assert abs(lhs - _dot(dw_dir, w.grad)) <= 2e-6 * _scale(cdir.detach(), yv)   # <conv_dw(x), y> = <dw, wgrad(x, y)>
    finally:
        ops.set_compute_dtype("fp32")


def test_bn_and_loss_head_full_resolution(dev):
    from vnet_tensorflow_amd import ops
    C = 16
    x = torch.randn(1, P, P, P, C, device=dev) * 3.0 + 1.5
    r = torch.randn(1, P, P, P, C, device=dev)
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev)
    y = ops.bn_act(x, gamma, beta, act=None, residual=r)
    s = (x + r).double().reshape(-1, C)
    var = s.var(0, unbiased=False)
    yd = y.double().reshape(-1, C)
    assert float((yd.mean(0) - beta.double()).abs().max()) < 1e-5                       # mean = beta
    ref_var = gamma.double() ** 2 * var / (var + 1e-3)
    assert float(((yd.var(0, unbiased=False) - ref_var) / ref_var).abs().max()) < 1e-5   # var = g^2 s^2/(s^2+eps)
    # softmax + Dice closed forms: uniform logits vs a label map with n_c voxels of class c
    K = 2
    lab = torch.zeros(1, P, P, P, 1, dtype=torch.int32, device=dev)
    lab[:, :32] = 1
    n1 = 32 * P * P
    n = [P ** 3 - n1, n1]
    loss, dice, sm, pred = ops.softmax_loss(torch.zeros(1, P, P, P, K, device=dev), lab, "sorensen", want_softmax=True)
    assert float((sm - 0.5).abs().max()) == 0.0
    ref = np.mean([(2.0 * n[c] / K + 1e-5) / (P ** 3 / K + n[c] + 1e-5) for c in range(K)])
    assert abs(float(loss) - (1.0 - ref)) < 1e-6
    big = torch.zeros(1, P, P, P, K, device=dev)
    big.scatter_(-1, lab.long(), 40.0)                                                   # a perfect, saturated prediction
    loss, _, _, pred = ops.softmax_loss(big, lab, "sorensen", want_pred=True)
    assert abs(float(loss)) < 1e-6 and torch.equal(pred.reshape(-1), lab.reshape(-1).long())


def test_network_c3_full_size_properties(dev):
    """Config C3 (128^3, 1 modality, 2 classes, full-width net), one fwd+loss+bwd:
    deterministic run to run (bit-identical logits, loss and gradients: no atomics on the path); duplicating the patch
    in the batch leaves train-mode batch-norm statistics, hence logits and loss, unchanged and doubles no gradient
    (mean over the batch): B=2 == B=1 to fp32 roundoff."""
    from vnet_tensorflow_amd import networks, ops, optim
    np.random.seed(7)
    net = networks.VNet(2, 0.0, 16, 4, (1, 2, 3, 3), 3, True, "prelu", device=dev)
    net.build((1, P, P, P, 1))
    flat = optim.FlatParams(net.named_parameters())
    x, lab = O.synthetic_batch(1, P, 1, 2, seed=1000)
    tx, tl = g(x, dev), g(lab, dev, torch.int32)

    def run(xb, lb):
        flat.zero_grad()
        logits = net.GetNetwork(xb)
        loss, _, _, _ = ops.softmax_loss(logits, lb, "sorensen")
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), float(loss.detach()), flat.grad.clone()

    l1, loss1, g1 = run(tx, tl)
    l1b, loss1b, g1b = run(tx, tl)
    assert torch.equal(l1, l1b) and loss1 == loss1b and torch.equal(g1, g1b)
    assert np.isfinite(loss1) and 0.0 < loss1 < 1.0 and bool(torch.isfinite(g1).all())
    l2, loss2, g2 = run(torch.cat((tx, tx)), torch.cat((tl, tl)))
    assert float((l2[0:1] - l1).abs().max()) < 2e-4 and float((l2[1:2] - l1).abs().max()) < 2e-4
    assert abs(loss2 - loss1) < 1e-6
    assert float((g2 - g1).norm() / g1.norm()) < 1e-3


def test_adam_full_parameter_vector(dev):
    """TF-form Adam over the full 43.9 M-parameter flat buffer against the formula (model.py:649, Appendix A)."""
    from vnet_tensorflow_amd import ops
    n = 43_940_000
    gen = torch.Generator(device=dev).manual_seed(3)
    p = torch.randn(n, device=dev, generator=gen)
    grad = torch.randn(n, device=dev, generator=gen) * 1e-2
    m = torch.randn(n, device=dev, generator=gen) * 1e-3
    v = torch.rand(n, device=dev, generator=gen) * 1e-4
    lr, b1, b2, eps, t, gs = 1e-2, 0.9, 0.999, 1e-8, 7, 0.125
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    gd = grad.double() * gs                                   # reference in float64 from the same fp32 inputs
    # TF's ApplyAdam form with the fp32 scalars the kernel receives: m += (g - m)(1 - beta1), v += (g^2 - v)(1 - beta2)
    omb1 = float(np.float32(1) - np.float32(b1)); omb2 = float(np.float32(1) - np.float32(b2))
    gd = (grad * np.float32(gs)).double()
    mr = m.double() + (gd - m.double()) * omb1
    vr = v.double() + (gd * gd - v.double()) * omb2
    step = float(np.float32(lr_t)) * mr / (vr.sqrt() + float(np.float32(eps)))
    pr = p.double() - step
    ops.adam_apply(p, grad, m, v, lr_t, b1, b2, eps, gs)
    # fp32 roundoff of m and v is amplified where sqrt(v) is tiny (steps up to O(1)): bound relative to |p| + |step|
    assert float(((p.double() - pr).abs() / (pr.abs() + step.abs() + 1e-3)).max()) < 5e-7
    assert float((m.double() - mr).abs().max()) < 1e-9 and float((v.double() - vr).abs().max()) < 1e-11

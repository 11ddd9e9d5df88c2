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
    (1, 32, 32, 64, 16, 16, 16),   # backward-data into two 16-channel destinations with TWO cout blocks per item: the pair straddles them
    # volumes exactly 8 wide (the 8^3 level): the narrow brick 4 x 8 x 8 -- lanes 8..15 of a column block are the plane two further
    (1, 8, 8, 8, 32, 0, 32),       # whole bricks, two chunks (the second runs negated), two cout blocks
    (2, 6, 10, 8, 16, 16, 48),     # ragged depth and height, two sources, batch 2
    (1, 3, 5, 8, 16, 0, 16),       # smaller than one brick
    (1, 8, 8, 8, 128, 0, 64),      # eight chunks split over four workgroups per item (partial slabs + reduce)
    (1, 9, 8, 8, 48, 0, 16),       # three chunks, a last brick with one plane
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


def test_x3_two_cout_blocks_per_item_is_bit_identical(dev, split3):
    """`X3_NB2` (two 16-cout blocks per item: the tile commit and its barriers paid once per pair) changes the schedule, not the sums."""
    ops = split3
    from vnet_tensorflow_amd import _lib
    torch.manual_seed(3)
    x = torch.randn(1, 32, 32, 32, 64, device=dev); w = torch.randn(5, 5, 5, 64, 64, device=dev) * 0.05; b = torch.randn(64, device=dev)
    ys = []
    try:
        for v in (0, 1):
            _lib.set_option("X3_NB2", v)
            ys.append(ops.conv(x, w, b, 5, 1))
    finally:
        _lib.set_option("X3_NB2", 1)
    assert torch.equal(ys[0], ys[1])


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


@pytest.mark.parametrize("W", [20, 8])
def test_x3_epilogue_statistics_accumulate_and_residual(dev, split3, W):
    """The batch-norm partial sums of the epilogue (rows per 2x8x16 brick; per 4x8x8 brick in a volume 8 wide), the residual in front of
    them, and y += conv."""
    ops = split3
    from vnet_tensorflow_amd import _lib
    L = _lib.lib()
    B, D, H, C, Co = 1, 5, 9, 32, 32
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, D, H, W, C)); w = rng.standard_normal((5, 5, 5, C, Co)) * 0.1
    b = rng.standard_normal(Co); r = rng.standard_normal((B, D, H, W, Co)); y_old = rng.standard_normal((B, D, H, W, Co))
    y_ref = O.conv_nd_fwd(x, w, 1) + b
    tx, tw, tb, tr = g(x, dev), g(w, dev), g(b, dev), g(r, dev)
    wp = ops.packed_weights(tw, ops.PACK_FWD_X3, 125, C, Co)
    rows = L.vnet_conv_x3_stats_rows(C, Co, B, D, H, W)
    assert rows == (3 * 2 * 2 if W == 20 else 2 * 2 * 1)
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
    from the packed filter image [k chunk][pair 63][n block][piece 3][64 lanes][8 k] and add them in float64."""
    ops = split3
    rng = np.random.default_rng(12)
    I = O = 16
    w = (rng.standard_normal((5, 5, 5, I, O)) * np.exp2(rng.integers(-30, 31, (5, 5, 5, I, O)))).astype(np.float32)
    w.reshape(-1)[:7] = [0.0, -0.0, 1.0, -1.0, 3.0e38, 1.0e-30, 2.0 ** -100]      # (not within 2^24 of the smallest normal: the last piece would be denormal)
    tw = g(w, dev)
    img = ops.packed_weights(tw, ops.PACK_FWD_X3, 125, I, O).cpu().numpy().view(np.uint16).reshape(63, 3, 64, 8)
    f = (img.astype(np.uint32) << 16).view(np.float32).astype(np.float64)        # bf16 -> float
    total = f.sum(1)                                                             # [pair][lane][8 k]
    w3 = w.reshape(125, I, O).astype(np.float64)
    for p in range(63):
        for hi in range(2):
            if p < 50:                                   # (dz, dz + 1) x dy x dx
                zp, r = divmod(p, 25); dx, dy = divmod(r, 5); dz = 2 * zp + hi; valid = True
            elif p < 60:                                 # plane 4: (dy, dy + 1) for dy = 0, 2
                dx, q = divmod(p - 50, 2); dz = 4; dy = 2 * q + hi; valid = True
            else:                                        # row (4, 4): (dx, dx + 1) for dx = 0, 2; (4, 4, 4) alone
                dz = dy = 4; dx = 2 * (p - 60) + hi; valid = not (p == 62 and hi)
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


# ---- round 6 (VERDICT r5 next #1 b, c): the gates for reporting fp32_split3 as the fp32 number ---------------------------------------
# Shape of the A/B: 16 x 32 x 64 voxels, 32 -> 32 channels = 256 (brick, cout block) work items for BOTH kernels, i.e. the accumulation
# structure of the bench sizes: every output is ONE chain of 4000 products (the fp32 MFMA kernel: 1000 v_mfma_f32_16x16x4_f32 in a row;
# f32x3: four waves x ~195 v_mfma_f32_16x16x32_bf16 each, met once in LDS).  On smaller volumes the fp32-MFMA planner splits K over up to
# ten partial slabs (plan_conv: nsplit x nz) and the slab-wise summation makes THAT launch more accurate than either kernel is at
# 128^3 -- measured on 6 x 16 x 32: `spread` forward 1.65e-7 (ten slabs) against 2.5e-7 (f32x3) -- which says something about
# split-K, not about the two arithmetics.
X3_AB_SHAPE = (1, 16, 32, 64, 32, 32)


def _adversarial_case(kind, rng):
    """(x [B,D,H,W,C], w [5,5,5,C,Co]) of operands chosen to hurt a split-operand product.
    raw:     intensities 0..255 with mean >> std, un-normalised (what a first 16-channel layer would see if the pipeline skipped
             its normalisation): the h piece carries almost everything, the information is in m and l;
    cancel:  alternating-sign filter on a smooth input: the 4000-term sum cancels to ~1e-3 of its terms' magnitude, so every
             absolute product error is amplified ~1000x in the result;
    spread:  magnitudes spread over 2^+-20 in both operands, independently per ELEMENT (the pieces' exponents are all over the place,
             also inside the 8 consecutive channels one lane feeds to an MFMA, whose products the bf16 instruction adds in a 24-bit
             window below the largest of them: profiles/r06_mfma_round_probe.txt);
    spread_vox: the same 2^+-20 spread per VOXEL of x and per (tap, cout) of w -- the 8 channels of a lane share a scale;
    tiny_m:  operands that are bf16-representable plus a 2^-17 perturbation: m is a single bit, l = 0."""
    B, D, H, W, C, Co = X3_AB_SHAPE
    if kind == "raw":
        x = np.clip(np.rint(180.0 + 6.0 * rng.standard_normal((B, D, H, W, C))), 0, 255)
        w = rng.standard_normal((5, 5, 5, C, Co)) * 0.02
    elif kind == "cancel":
        z, y, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
        smooth = 5.0 + np.sin(0.05 * z + 0.03 * y + 0.02 * xx)
        x = smooth[None, ..., None] * (1.0 + 1e-3 * rng.standard_normal((B, D, H, W, C)))
        tz, ty, tx = np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij")
        sign = np.where((tz + ty + tx) % 2 == 0, 1.0, -1.0)[..., None, None]
        w = sign * (0.1 + 1e-4 * rng.standard_normal((5, 5, 5, C, Co)))
        w -= w.mean(axis=(0, 1, 2), keepdims=True) * 0.999           # taps of a (ci, co) pair nearly sum to zero
    elif kind == "spread":
        x = rng.standard_normal((B, D, H, W, C)) * np.exp2(rng.integers(-20, 21, (B, D, H, W, C)))
        w = rng.standard_normal((5, 5, 5, C, Co)) * np.exp2(rng.integers(-20, 21, (5, 5, 5, C, Co)))
    elif kind == "spread_vox":
        x = rng.standard_normal((B, D, H, W, C)) * np.exp2(rng.integers(-20, 21, (B, D, H, W, 1)))
        w = rng.standard_normal((5, 5, 5, C, Co)) * np.exp2(rng.integers(-20, 21, (5, 5, 5, 1, Co)))
    else:
        def near_bf16(a):
            u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
            return u.view(np.float32).astype(np.float64) * (1.0 + np.exp2(-17) * rng.integers(-1, 2, a.shape))
        x = near_bf16(rng.standard_normal((B, D, H, W, C)))
        w = near_bf16(rng.standard_normal((5, 5, 5, C, Co)) * 0.1)
    f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    return f(x), f(w)


@pytest.mark.parametrize("kind", ["raw", "cancel", "spread", "spread_vox", "tiny_m"])
def test_x3_not_worse_than_fp32_mfma_on_adversarial_operands(dev, kind):
    """A/B against float64 (VERDICT r5 next #1b): the SAME operands through the fp32-MFMA kernels (v_mfma_f32_16x16x4_f32) and through
    the f32x3 kernels, forward + backward-data + filter gradient; the f32x3 error (rel-L2 vs the numpy-fp64 oracle) must not exceed
    1.25 x the fp32 MFMA's on any of them (measured ratios: profiles/r06_x3_adversarial.txt).  Both are also held to an absolute
    2e-6 x the conditioning of the case: the `cancel` sum loses three digits by construction -- its yardstick is sum |x||w|, not
    |sum x w|."""
    from vnet_tensorflow_amd import _lib, ops
    rng = np.random.default_rng({"raw": 1, "cancel": 2, "spread": 3, "tiny_m": 4, "spread_vox": 5}[kind])
    x, w = _adversarial_case(kind, rng)
    B, D, H, W, C, Co = X3_AB_SHAPE
    L = _lib.lib()
    # the premise of the shape: neither kernel splits K (no partial slabs, no reduce launch)
    assert L.vnet_conv_ws_bytes(5, 0, 1, 0, C, Co, B, D, H, W) == 0 and L.vnet_conv_x3_ws_bytes(C, Co, B, D, H, W) == 0
    y_ref = O.conv_nd_fwd(x, w, 1)
    dy = rng.standard_normal(y_ref.shape)
    if kind == "spread":
        dy = dy * np.exp2(rng.integers(-20, 21, dy.shape))
    elif kind == "spread_vox":
        dy = dy * np.exp2(rng.integers(-20, 21, dy.shape[:-1] + (1,)))
    dy = dy.astype(np.float32).astype(np.float64)
    dx_ref, dw_ref = O.conv_nd_bwd(x, w, dy, 1)
    # conditioning: |x| * |w| against |x * w| (1 for benign operands)
    cond = float(np.linalg.norm(O.conv_nd_fwd(np.abs(x), np.abs(w), 1)) / np.linalg.norm(y_ref))
    errs = {}
    assert L.vnet_conv_x3_ok(C, 0, Co, 0, B, D, H, W) == 1            # the product's own dispatch takes the f32x3 convolution here
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        ops._X3["force"] = mode == "fp32_split3"      # (the filter gradient asks for >= 4 bricks per workgroup before it pays: force it, 2 here)
        try:
            tx, tw = g(x, dev).requires_grad_(True), g(w, dev).requires_grad_(True)
            ops.profile_start()
            y = ops.conv(tx, tw, None, 5, 1)
            y.backward(g(dy, dev))
            recs = ops.profile_stop()
            n3 = sum(1 for r in recs if r[0].startswith(("conv-x3", "wgrad-x3")))
            assert n3 == (3 if mode == "fp32_split3" else 0), [r[0] for r in recs]
            errs[mode] = (rel_l2(y.detach().cpu().numpy(), y_ref), rel_l2(tx.grad.cpu().numpy(), dx_ref), rel_l2(tw.grad.cpu().numpy(), dw_ref))
        finally:
            ops._X3["force"] = False
            ops.set_compute_dtype("fp32")
    print("adversarial %-10s cond %.1e  fp32 MFMA fwd/dx/dw %.2e %.2e %.2e | f32x3 %.2e %.2e %.2e | ratio %.2f %.2f %.2f" %
          ((kind, cond) + errs["fp32"] + errs["fp32_split3"] + tuple(b / a for a, b in zip(errs["fp32"], errs["fp32_split3"]))))
    bad = [(what, e3, e32, e3 / e32) for what, e32, e3 in zip(("fwd", "dx", "dw"), errs["fp32"], errs["fp32_split3"]) if e3 > 1.25 * e32 + 1e-9]
    assert not bad, (kind, bad)
    assert errs["fp32_split3"][0] <= 2e-6 * max(cond, 1.0), (errs, cond)


def test_x3_non_finite_operands(dev, split3):
    """Non-finite semantics of the split (VERDICT r5 next #1c; csrc/x3_pack.h:7-12, INTEGRATION.md "fp32_split3 and non-finite values").
    h = RNE_bf16(x) is +-Inf for x = +-Inf AND for finite |x| >= 0x7F7F8000 (3.3895e38: the values that round up to Inf in bf16);
    m = x - h is then Inf - Inf = NaN.  So every output whose 5^3 receptive field holds such an input is NaN -- where the fp32 MFMA
    kernels give +-Inf (or NaN when +Inf and -Inf products meet, or a finite number for a finite huge x times a small weight).
    Pinned here: (1) exactly the outputs that see a poisoned input are non-finite, and they are NaN; (2) every other output is
    BIT-identical to the clean run (a NaN does not leak through the workgroup's shared tiles / reductions); (3) the largest
    magnitude that does NOT round to Inf (0x7F7F7FFF) is still split exactly and gives a finite, correct result."""
    ops = split3
    rng = np.random.default_rng(21)
    B, D, H, W, C, Co = 1, 6, 16, 32, 16, 16
    x = rng.standard_normal((B, D, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((5, 5, 5, C, Co)) * 0.1).astype(np.float32)
    w[w == 0] = 0.1
    y_clean = ops.conv(g(x, dev), g(w, dev), None, 5, 1).cpu().numpy()
    poison = {(0, 1, 3, 4, 5): np.inf, (0, 4, 12, 20, 0): -np.inf, (0, 2, 8, 30, 9): np.float32(3.4e38), (0, 5, 0, 0, 15): np.nan}
    xp = x.copy()
    for k, v in poison.items():
        xp[k] = v
    y = ops.conv(g(xp, dev), g(w, dev), None, 5, 1).cpu().numpy()
    seen = np.zeros((B, D, H, W), dtype=bool)
    for (b, z, yy, xx, c) in poison:
        seen[b, max(0, z - 2):z + 3, max(0, yy - 2):yy + 3, max(0, xx - 2):xx + 3] = True
    assert np.isnan(y[seen]).all()                                   # NaN (not +-Inf) in every channel of every voxel that sees one
    assert np.array_equal(y[~seen].view(np.uint32), y_clean[~seen].view(np.uint32))
    # the fp32 MFMA kernels on the same operands: non-finite in the same voxels (Inf or NaN), finite elsewhere
    ops.set_compute_dtype("fp32"); ops._X3["force"] = False
    y32 = ops.conv(g(xp, dev), g(w, dev), None, 5, 1).cpu().numpy()
    ops.set_compute_dtype("fp32_split3"); ops._X3["force"] = True
    assert np.isfinite(y32[~seen]).all() and np.isinf(y32[seen]).any()
    # (3) the largest finite value whose bf16 rounding is finite
    big = np.array([0x7F7F7FFF], dtype=np.uint32).view(np.float32)[0]
    xb = np.zeros_like(x); xb[0, 3, 8, 16, 2] = big
    wb = np.zeros_like(w); wb[2, 2, 2, 2, :] = np.float32(2.0 ** -10) * np.arange(1, Co + 1)
    yb = ops.conv(g(xb, dev), g(wb, dev), None, 5, 1).cpu().numpy()
    assert np.isfinite(yb).all()
    # (w = k 2^-10 is a single piece; the three products h w, m w, l w are exact and meet in two fp32 additions: within one ulp)
    np.testing.assert_allclose(yb[0, 3, 8, 16, :].astype(np.float64), np.float64(big) * wb[2, 2, 2, 2, :].astype(np.float64), rtol=1.2e-7)


def test_filter_gradient_long_chains_carry_no_offset(dev):
    """Round 6: over the 10^5..10^6 voxels of a filter gradient the bf16 matrix instruction's one-sided accumulation error adds up
    to an OFFSET -- before the sign alternation of wgrad5_x3_kernel: mean error -7.9e-7 of rms |dw| on this very case (N(0,1)
    operands, 32 x 64 x 128 voxels, 16 -> 16; 256 workgroups x 4 bricks), rel-L2 9.3e-7 against the fp32 MFMA's 5.7e-7, and growing
    with the volume (profiles/r06_x3_wgrad_taps.txt).  The per-kernel tests above use volumes too small to see it and the 128^3
    crop test feeds a gradient supported on a 10^3 block.  Pinned here for BOTH fp32 modes: no offset (|mean error| <= 2.5e-7 of rms
    |dw|: measured +1.0e-7 / +7e-10), f32x3 rel-L2 <= 1.25 x the fp32 MFMA's (measured 0.86 x), both <= 1e-6."""
    from vnet_tensorflow_amd import ops
    D, H, W, C, Co = 32, 64, 128, 16, 16
    rng = np.random.default_rng(3)
    f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    x, dy = f(rng.standard_normal((1, D, H, W, C))), f(rng.standard_normal((1, D, H, W, Co)))
    w = f(rng.standard_normal((5, 5, 5, C, Co)) * 0.1)
    _, dw_ref = O.conv_nd_bwd(x, w, dy, 1, need_dx=False)
    rms = float(np.sqrt((dw_ref ** 2).mean()))
    res = {}
    for mode in ("fp32", "fp32_split3"):
        ops.set_compute_dtype(mode)
        try:
            tx, tw = g(x, dev), g(w, dev).requires_grad_(True)
            ops.profile_start()
            ops.conv(tx, tw, None, 5, 1).backward(g(dy, dev))
            recs = ops.profile_stop()
            assert any(r[0].startswith("wgrad-x3") for r in recs) == (mode == "fp32_split3"), [r[0] for r in recs]   # the product's own dispatch
            got = tw.grad.cpu().numpy().astype(np.float64)
            res[mode] = (rel_l2(got, dw_ref), float((got - dw_ref).mean()) / rms)
        finally:
            ops.set_compute_dtype("fp32")
    print("long-chain dw: fp32 MFMA rel-L2 %.2e offset %+.2e | f32x3 rel-L2 %.2e offset %+.2e" % (res["fp32"] + res["fp32_split3"]))
    for mode, (e, off) in res.items():
        assert e < 1e-6 and abs(off) < 2.5e-7, (mode, e, off)
    assert res["fp32_split3"][0] <= 1.25 * res["fp32"][0], res

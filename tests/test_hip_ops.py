"""-m gpu: every HIP kernel family, through the C ABI (ctypes via vnet_tensorflow_amd.ops), against
the numpy-fp64 oracle on the same seeded inputs.  Tolerances are fp32-roundoff class (stated per test)."""
import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.util import g, check_close

pytestmark = pytest.mark.gpu


def _conv_case(dev, B, D, H, W, C0, C1, Co, ks, stride, seed):
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((B, D, H, W, C0))
    x1 = rng.standard_normal((B, D, H, W, C1)) if C1 else None
    w = rng.standard_normal((ks, ks, ks, C0 + C1, Co)) * 0.1
    b = rng.standard_normal(Co)
    xcat = x0 if x1 is None else np.concatenate((x0, x1), -1)
    y_ref = O.conv_nd_fwd(xcat, w, stride) + b
    dy = rng.standard_normal(y_ref.shape)
    dx_ref, dw_ref = O.conv_nd_bwd(xcat, w, dy, stride)
    tx0 = g(x0, dev).requires_grad_(True)
    tx1 = g(x1, dev).requires_grad_(True) if C1 else None
    tw, tb = g(w, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
    y = ops.conv(tx0, tw, tb, ks, stride, x1=tx1)
    tag = "conv k%d s%d [%d,%d,%d,%d] %d+%d->%d" % (ks, stride, B, D, H, W, C0, C1, Co)
    check_close(tag + " fwd", y, y_ref, 2e-6)
    y.backward(g(dy, dev))
    check_close(tag + " dx0", tx0.grad, dx_ref[..., :C0], 2e-6)
    if C1:
        check_close(tag + " dx1", tx1.grad, dx_ref[..., C0:], 2e-6)
    check_close(tag + " dw", tw.grad, dw_ref, 2e-6)
    # a bias gradient is a sum of ~N(0,1) terms that cancel: fp32 round-off is relative to sum|dy|, not to the result
    check_close(tag + " db", tb.grad, dy.reshape(-1, Co).sum(0), 2e-6, atol=1e-7 * float(np.abs(dy).reshape(-1, Co).sum(0).max()))


@pytest.mark.parametrize("shape", [
    (1, 8, 16, 32, 16, 0, 16),     # wide brick, NS=1
    (2, 5, 9, 17, 16, 16, 16),     # ragged dims, two-source (concat), batch 2
    (1, 8, 8, 16, 32, 0, 32),      # NS=2, two chunks
    (1, 4, 8, 16, 64, 0, 64),      # NS=4, four chunks
    (1, 8, 8, 8, 32, 32, 32),      # cube brick (W<16), split-K
    (1, 4, 4, 4, 128, 0, 128),     # tiny volume, many chunks -> split-K, 2 cout blocks
    (1, 2, 2, 2, 64, 0, 64),       # 2^3 volume (bottom of config C1)
    (1, 6, 7, 9, 4, 4, 8),         # narrow channels: zero-padded chunk, Cout<16
    (1, 6, 6, 18, 3, 0, 16),       # Cin not a multiple of 4: scalar gather path
    (1, 4, 6, 16, 16, 0, 5),       # Cout not a multiple of 4: scalar scatter path
])
def test_conv5(dev, shape):
    _conv_case(dev, *shape, ks=5, stride=1, seed=sum(shape))


@pytest.mark.parametrize("shape", [
    (1, 8, 16, 32, 16, 0, 32),
    (2, 4, 8, 16, 32, 0, 64),
    (1, 8, 8, 8, 64, 0, 128),
    (1, 6, 10, 14, 8, 0, 16),
    (1, 5, 7, 9, 4, 0, 8),         # odd dims: SAME pads one voxel on the high side
    (1, 2, 2, 2, 128, 0, 256),
])
def test_down_conv(dev, shape):
    _conv_case(dev, *shape, ks=2, stride=2, seed=sum(shape))


@pytest.mark.parametrize("shape", [
    # B, d, h, w (coarse), Cin, Cout, (out dims)
    (1, 4, 8, 16, 32, 16, None),
    (2, 4, 4, 8, 64, 32, None),
    (1, 8, 8, 8, 128, 64, None),
    (1, 1, 1, 1, 256, 128, None),
    (1, 3, 4, 5, 8, 4, (5, 7, 9)),   # odd output shape (skip tensor of an odd level)
])
def test_up_conv(dev, shape):
    from vnet_tensorflow_amd import ops
    B, d, h, w_, Ci, Co, outsp = shape
    rng = np.random.default_rng(sum(shape[:6]))
    outsp = outsp or (2 * d, 2 * h, 2 * w_)
    x = rng.standard_normal((B, d, h, w_, Ci))
    w = rng.standard_normal((2, 2, 2, Co, Ci)) * 0.2
    b = rng.standard_normal(Co)
    X, Wv, Bv = O.Var(x), O.Var(w), O.Var(b)
    y = O.deconvolution(X, Wv, Bv, outsp, 2)
    dy = rng.standard_normal(y.v.shape)
    O.backward(y, dy)
    tx, tw, tb = (g(a, dev).requires_grad_(True) for a in (x, w, b))
    ty = ops.conv_transpose2(tx, tw, tb, outsp)
    tag = "upconv %s" % (shape,)
    check_close(tag + " fwd", ty, y.v, 2e-6)
    ty.backward(g(dy, dev))
    check_close(tag + " dx", tx.grad, X.g, 2e-6)
    check_close(tag + " dw", tw.grad, Wv.g, 2e-6)
    check_close(tag + " db", tb.grad, Bv.g, 2e-6)


def _up_case(dev, shape, seed):
    from vnet_tensorflow_amd import ops
    B, d, h, w_, Ci, Co, outsp = shape
    rng = np.random.default_rng(seed)
    outsp = outsp or (2 * d, 2 * h, 2 * w_)
    x = rng.standard_normal((B, d, h, w_, Ci))
    w = rng.standard_normal((2, 2, 2, Co, Ci)) * 0.2
    b = rng.standard_normal(Co)
    X, Wv, Bv = O.Var(x), O.Var(w), O.Var(b)
    y = O.deconvolution(X, Wv, Bv, outsp, 2)
    dy = rng.standard_normal(y.v.shape)
    O.backward(y, dy)
    tx, tw, tb = (g(a, dev).requires_grad_(True) for a in (x, w, b))
    ty = ops.conv_transpose2(tx, tw, tb, outsp)
    tag = "upconv %s" % (shape,)
    check_close(tag + " fwd", ty, y.v, 2e-6)
    ty.backward(g(dy, dev))
    check_close(tag + " dx", tx.grad, X.g, 2e-6)
    check_close(tag + " dw", tw.grad, Wv.g, 2e-6)
    check_close(tag + " db", tb.grad, Bv.g, 2e-6, atol=1e-7 * float(np.abs(dy).reshape(-1, Co).sum(0).max()))


def test_conv_family_random_shapes(dev):
    """Property test: the three convolution kernels of the path on RANDOM small problems -- odd and
    unit extents, batch 1..3, channel counts that are no multiple of 4 or 16, a second (concat) source or not --
    forward, backward-data, filter and bias gradient against the fp64 oracle, fp32-roundoff tolerance.  hypothesis
    draws the cases (derandomised: the same 40 cases on every run) and shrinks a failing one to its smallest form."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    dim = st.integers(1, 11)
    ch = st.one_of(st.integers(1, 24), st.sampled_from([16, 32, 48, 64]))

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(kind=st.sampled_from(["k5", "k5", "down", "up"]), B=st.integers(1, 3), D=dim, H=dim, W=st.integers(1, 20),
           C0=ch, C1=st.one_of(st.just(0), ch), Co=ch, odd=st.booleans(), seed=st.integers(0, 10 ** 6))
    def run(kind, B, D, H, W, C0, C1, Co, odd, seed):
        if kind == "k5":
            _conv_case(dev, B, D, H, W, C0, C1, Co, ks=5, stride=1, seed=seed)
        elif kind == "down":
            _conv_case(dev, B, D, H, W, C0, 0, Co, ks=2, stride=2, seed=seed)
        else:       # transposed conv onto a skip tensor of even or odd extent (output_shape = tf.shape(skip))
            out = tuple(2 * v - (1 if odd and v > 1 else 0) for v in (D, H, W))
            _up_case(dev, (B, D, H, W, C0, Co, out), seed)

    run()


@pytest.mark.parametrize("C,act,res,tile", [
    (16, "prelu", False, False), (16, "prelu", True, False), (32, "relu", True, False), (64, None, False, False),
    (256, "prelu", True, False), (16, None, False, True), (2, None, False, False), (5, "lrelu", True, False),
    (12, "prelu", False, False), (8, "prelu", False, True),
])
def test_bn_act(dev, C, act, res, tile):
    _bn_act_case(dev, C, act, res, tile, (2, 5, 6, 7), C + 7 * bool(res))


def test_bn_act_random_cases(dev):
    """Property test: train-mode batch-norm (+ residual | + tile) + activation on random row counts and channel counts
    (vector path: C/4 a power of two; row path: C <= 8; generic path: everything else), forward, all gradients and the
    moving-average update against the fp64 oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(C=st.one_of(st.integers(1, 40), st.sampled_from([16, 32, 64, 128, 256])), act=st.sampled_from([None, "relu", "prelu", "lrelu"]),
           extra=st.sampled_from(["none", "res", "tile"]), B=st.integers(1, 3), D=st.integers(1, 7), H=st.integers(1, 7),
           W=st.integers(2, 19), seed=st.integers(0, 10 ** 6))
    def run(C, act, extra, B, D, H, W, seed):
        if B * D * H * W < 4:
            W += 4                      # the batch variance of fewer than a handful of rows is all round-off
        _bn_act_case(dev, C, act, extra == "res", extra == "tile", (B, D, H, W), seed)

    run()


@pytest.mark.parametrize("C", [10, 37, 300, 1021])
def test_bn_stats_generic_path_is_deterministic(dev, C):
    """Channel counts that are neither a multiple of 4 nor <= 8 take the generic statistics kernel (thread = (row group, channel)
    for C <= 256, one row per block pass above): right against fp64, and bit-identical from launch to launch (it used LDS float
    atomics before round 3)."""
    from vnet_tensorflow_amd import _lib, ops
    L = _lib.lib()
    rng = np.random.default_rng(C)
    M = 5000
    x = (rng.standard_normal((M, C)) * 2.0 + 0.5).astype(np.float32)
    tx = torch.from_numpy(x).to(dev)
    nb = L.vnet_bn_ws_bytes(C)
    outs = []
    for _ in range(3):
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        rc = L.vnet_bn_stats(ops._ptr(tx), None, 0, M, C, 1e-3, 0.99, ops._ptr(mean), ops._ptr(invstd), None, None, ops._ptr(ws), nb, ops._stream())
        assert rc == 0
        outs.append((mean.cpu().numpy().copy(), invstd.cpu().numpy().copy()))
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(outs[0][0], x64.mean(0), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(outs[0][1], 1.0 / np.sqrt(x64.var(0) + 1e-3), rtol=5e-6)
    for m, i in outs[1:]:
        assert np.array_equal(m, outs[0][0]) and np.array_equal(i, outs[0][1])


def _bn_act_case(dev, C, act, res, tile, shp, seed):
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shp + ((1,) if tile else (C,))) * 3.0 + 1.5
    r = rng.standard_normal(shp + (C,)) if res else None
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C)
    alpha = rng.uniform(0.05, 0.3, C)
    X, G_, B_, A_ = O.Var(x), O.Var(gamma), O.Var(beta), O.Var(alpha)
    s = O.tile_channels(X, C) if tile else X
    R = O.Var(r) if res else None
    if res:
        s = O.add(s, R)
    st = []
    y = O.batch_norm_train(s, G_, B_, stats_out=st)
    y = O.prelu(y, A_) if act == "prelu" else O.relu(y) if act == "relu" else O.leaky_relu(y) if act == "lrelu" else y
    dy = rng.standard_normal(y.v.shape)
    O.backward(y, dy)
    tx = g(x, dev).requires_grad_(True)
    tr = g(r, dev).requires_grad_(True) if res else None
    tg, tb, ta = (g(a, dev).requires_grad_(True) for a in (gamma, beta, alpha))
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ty = ops.bn_act(tx, tg, tb, act, ta if act == "prelu" else None, tr, tile, mm, mv)
    tag = "bn_act C%d %s res%d tile%d" % (C, act, res, tile)
    check_close(tag + " fwd", ty, y.v, 5e-6)
    ty.backward(g(dy, dev))
    check_close(tag + " dx", tx.grad, X.g, 5e-5, atol=1e-5)
    if res:
        check_close(tag + " dr", tr.grad, R.g, 5e-5, atol=1e-5)
    check_close(tag + " dgamma", tg.grad, G_.g, 2e-5)
    check_close(tag + " dbeta", tb.grad, B_.g, 2e-5)
    if act == "prelu":
        check_close(tag + " dalpha", ta.grad, A_.g, 2e-5)
    mu, var = st[0]
    check_close(tag + " moving_mean", mm, 0.01 * mu, 1e-5, atol=1e-7)
    check_close(tag + " moving_var", mv, 0.99 + 0.01 * var, 1e-5)


@pytest.mark.parametrize("kind,C,act", [(0, 16, "prelu"), (0, 32, "relu"), (0, 5, None), (1, 16, "prelu"), (1, 64, "lrelu"), (1, 6, None)])
def test_bn_chain(dev, kind, C, act):
    """The decoder's batch-norm chains evaluated in closed form as ONE normalisation of x (ops.bn_chain) against the
    oracle's layer-by-layer graph: kind 0 = act(BN3(BN1(x) + BN2(BN1(x)))), kind 1 = act(BNb(x + BNa(x)));
    gammas of both signs and small magnitude, a per-channel variance spread over 3 decades (so var is not >> eps)."""
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(100 * kind + C)
    shp = (2, 5, 6, 7)
    scale = np.exp(rng.uniform(np.log(0.02), np.log(20.0), C))
    x = rng.standard_normal(shp + (C,)) * scale * (1.0 + rng.standard_normal(C) * 0.5)     # |mean| stays ~std: fp32 BN is conditioned by |mean|/std
    x = x.astype(np.float32).astype(np.float64)
    nl = 3 if kind == 0 else 2
    gam = [rng.uniform(0.3, 1.5, C) * rng.choice([-1.0, 1.0], C) for _ in range(nl)]
    bet = [rng.standard_normal(C) for _ in range(nl)]
    alpha = rng.uniform(0.05, 0.3, C)
    X, A_ = O.Var(x), O.Var(alpha)
    G_ = [O.Var(v) for v in gam]
    B_ = [O.Var(v) for v in bet]
    st = [[], [], []]
    if kind == 0:
        y1 = O.batch_norm_train(X, G_[0], B_[0], stats_out=st[0])
        y2 = O.batch_norm_train(y1, G_[1], B_[1], stats_out=st[1])
        y = O.batch_norm_train(O.add(y1, y2), G_[2], B_[2], stats_out=st[2])
    else:
        r = O.batch_norm_train(X, G_[0], B_[0], stats_out=st[0])
        y = O.batch_norm_train(O.add(X, r), G_[1], B_[1], stats_out=st[1])
    y = O.prelu(y, A_) if act == "prelu" else O.relu(y) if act == "relu" else O.leaky_relu(y) if act == "lrelu" else y
    dy = rng.standard_normal(y.v.shape)
    O.backward(y, dy)
    tx = g(x, dev).requires_grad_(True)
    tg = [g(v, dev).requires_grad_(True) for v in gam]
    tb = [g(v, dev).requires_grad_(True) for v in bet]
    ta = g(alpha, dev).requires_grad_(True)
    mov = []
    for _ in range(3):
        mov += [torch.zeros(C, device=dev), torch.ones(C, device=dev)]
    if kind == 1:
        mov[4] = mov[5] = None
    ty = ops.bn_chain(tx, kind, act, ta if act == "prelu" else None, tg[0], tb[0], tg[1], tb[1],
                      tg[2] if kind == 0 else None, tb[2] if kind == 0 else None, tuple(mov))
    tag = "bn_chain kind%d C%d %s" % (kind, C, act)
    check_close(tag + " fwd", ty, y.v, 1e-5)
    ty.backward(g(dy, dev))
    check_close(tag + " dx", tx.grad, X.g, 5e-5, atol=1e-5)
    for k in range(nl):
        check_close(tag + " dgamma%d" % k, tg[k].grad, G_[k].g, 5e-5, atol=1e-5)
        # the inner layers' beta gradients are analytically zero (the oracle holds fp64 roundoff there)
        check_close(tag + " dbeta%d" % k, tb[k].grad, B_[k].g, 2e-5, atol=1e-6)
    if act == "prelu":
        check_close(tag + " dalpha", ta.grad, A_.g, 2e-5)
    for k in range(nl):
        mu, var = st[k][0]
        check_close(tag + " moving_mean%d" % k, mov[2 * k], 0.01 * mu, 1e-5, atol=1e-7)
        check_close(tag + " moving_var%d" % k, mov[2 * k + 1], 0.99 + 0.01 * var, 1e-5)


@pytest.mark.parametrize("direct", [True, False])
@pytest.mark.parametrize("shape", [(1, 8, 16, 32, 16), (2, 5, 9, 17, 16), (1, 6, 7, 9, 4), (1, 8, 8, 8, 8), (1, 12, 13, 140, 16), (1, 9, 12, 70, 8)])
def test_input_block(dev, shape, direct):
    """conv5^3(BN(tile(img))) through the un-tiled folded form (csrc/input_block.hip) == the oracle's tiled graph, including the
    gradients that reach the input BN's gamma/beta through BOTH the conv and a residual use of x.  direct (round 6): packed fp32 FMAs
    straight from the image for 8 / 16 channels (12 x 13 x 140 has bricks on every face, two ragged axes AND an interior brick, whose
    indicator channel is a constant); otherwise rounds 1-5's x-im2col + 5x5x1 MFMA kernels (what 4 channels still take)."""
    from vnet_tensorflow_amd import ops
    B, D, H, W, C = shape
    prev = ops._FUSE["input_direct"]
    ops._FUSE["input_direct"] = direct
    try:
        _input_block_case(dev, shape, direct)
    finally:
        ops._FUSE["input_direct"] = prev


def _input_block_case(dev, shape, direct):
    from vnet_tensorflow_amd import ops
    B, D, H, W, C = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.standard_normal((B, D, H, W, 1)) * 40 + 120
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C)
    w = rng.standard_normal((5, 5, 5, C, C)) * 0.1
    b = rng.standard_normal(C)
    IMG, G_, B_, W_, Bi = O.Var(img), O.Var(gamma), O.Var(beta), O.Var(w), O.Var(b)
    x = O.batch_norm_train(O.tile_channels(IMG, C), G_, B_)
    y = O.add(O.convolution(x, W_, Bi, 1), x)            # conv + residual, like the first encoder block
    dy = rng.standard_normal(y.v.shape)
    O.backward(y, dy)
    timg = g(img, dev)
    tg, tb, tw, tbi = (g(a, dev).requires_grad_(True) for a in (gamma, beta, w, b))
    tx, mean, invstd = ops.bn_act(timg, tg, tb, None, None, None, True, None, None, want_stats=True)
    ops.profile_start()
    ty = ops.input_conv(timg, tg, tb, mean, invstd, tw, tbi) + tx
    tag = "input block %s" % (shape,)
    check_close(tag + " fwd", ty, y.v, 5e-6)
    ty.backward(g(dy, dev))
    recs = ops.profile_stop()
    assert sum(1 for r in recs if r[0].startswith("input-")) == (2 if direct and C in (8, 16) else 0), [r[0] for r in recs]
    check_close(tag + " dw", tw.grad, W_.g, 1e-5)
    check_close(tag + " db", tbi.grad, Bi.g, 1e-5)
    check_close(tag + " dgamma", tg.grad, G_.g, 5e-5)
    check_close(tag + " dbeta", tb.grad, B_.g, 5e-5)


def test_input_block_direct_statistics(dev):
    """The direct input convolution's epilogue: batch-norm partial sums of y (+ residual), one row per 4 x 4 x 64 brick."""
    from vnet_tensorflow_amd import _lib, ops
    L = _lib.lib()
    B, D, H, W, C = 2, 6, 9, 70, 16
    rng = np.random.default_rng(11)
    img = g(rng.standard_normal((B, D, H, W, 1)) * 40 + 120, dev)
    tg, tb, tw, tbi = (g(a, dev) for a in (rng.uniform(0.5, 1.5, C), rng.standard_normal(C), rng.standard_normal((5, 5, 5, C, C)) * 0.1, rng.standard_normal(C)))
    tx, mean, invstd = ops.bn_act(img, tg, tb, None, None, None, True, None, None, want_stats=True)
    y = ops.input_conv(img, tg, tb, mean, invstd, tw, tbi, bn_stats=True, bn_residual=tx)
    st = getattr(y, "_vnet_stats", None)
    assert st is not None and st.rows == L.vnet_input_conv_direct_stats_rows(B, D, H, W) == 2 * 2 * 3 * 2
    prev = ops._FUSE["input_direct"]
    ops._FUSE["input_direct"] = False
    try:
        y_old = ops.input_conv(img, tg, tb, mean, invstd, tw, tbi)
    finally:
        ops._FUSE["input_direct"] = prev
    check_close("direct vs 5x5x1 MFMA form", y, y_old.cpu().numpy().astype(np.float64), 2e-6)
    v = (y.double() + tx.double()).reshape(-1, C).cpu().numpy()
    s = st.partial.double().cpu().numpy()
    np.testing.assert_allclose(s[:, :C].sum(0), v.sum(0), rtol=0, atol=2e-5 * np.abs(v).sum(0).max())
    np.testing.assert_allclose(s[:, C:].sum(0), (v * v).sum(0), rtol=2e-6)


@pytest.mark.parametrize("C,K", [(16, 2), (16, 5), (4, 3), (8, 8), (6, 2), (5, 3), (12, 4), (24, 1)])
def test_head(dev, C, K):
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(C * K)
    x = rng.standard_normal((2, 4, 5, 6, C))
    w = rng.standard_normal((1, 1, 1, C, K))
    b = rng.standard_normal(K)
    y_ref = x @ w[0, 0, 0] + b
    dy = rng.standard_normal(y_ref.shape)
    tx, tw, tb = (g(a, dev).requires_grad_(True) for a in (x, w, b))
    y = ops.head_conv(tx, tw, tb)
    check_close("head fwd", y, y_ref, 2e-6)
    y.backward(g(dy, dev))
    check_close("head dx", tx.grad, dy @ w[0, 0, 0].T, 2e-6)
    check_close("head dw", tw.grad, (x.reshape(-1, C).T @ dy.reshape(-1, K)).reshape(w.shape), 2e-6)
    check_close("head db", tb.grad, dy.reshape(-1, K).sum(0), 2e-6)


LOSSES = ["sorensen", "jaccard", "weighted_sorensen", "weighted_jaccard", "xent", "weighted_xent",
          "mixed_sorensen", "mixed_weighted_sorensen", "mixed_jaccard", "mixed_weighted_jaccard"]


@pytest.mark.parametrize("loss_name", LOSSES)
@pytest.mark.parametrize("B,K", [(2, 3), (1, 2), (3, 5)])
def test_softmax_loss(dev, loss_name, B, K):
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(B * 10 + K)
    z = rng.standard_normal((B, 6, 7, 9, K)) * 2.0
    lab = rng.integers(0, K, size=(B, 6, 7, 9, 1)).astype(np.int32)
    wts = list(rng.uniform(0.1, 1.0, K))
    Z = O.Var(z)
    loss, sm = O.loss_head(Z, lab, loss_name, wts, 0.7)
    O.backward(loss, 1.7)
    tz = g(z, dev).requires_grad_(True)
    tl, _, tsm, tpred = ops.softmax_loss(tz, g(lab, dev, torch.int32), loss_name, wts, 0.7, want_softmax=True, want_pred=True)
    check_close(loss_name + " loss", tl, loss.v, 2e-6)
    check_close(loss_name + " softmax", tsm, sm.v, 2e-6)
    assert (tpred.cpu().numpy() == O.argmax_pred(z)).all()
    (tl * 1.7).backward()
    check_close(loss_name + " dlogits", tz.grad, Z.g, 1e-5)


def test_softmax_loss_random_cases(dev):
    """Property test: the fused softmax + one-hot + Dice / cross-entropy head on random (B, volume, K in 1..8, loss name,
    weights, alpha) incl. labels OUTSIDE [0, K) (tf.one_hot gives an all-zero row): loss, softmax, argmax and
    dlogits against the fp64 oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from vnet_tensorflow_amd import ops

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(loss_name=st.sampled_from(LOSSES), B=st.integers(1, 3), K=st.integers(1, 8), D=st.integers(1, 9), H=st.integers(1, 9),
           W=st.integers(1, 33), stray=st.booleans(), alpha=st.floats(0.0, 2.0), seed=st.integers(0, 10 ** 6))
    def run(loss_name, B, K, D, H, W, stray, alpha, seed):
        rng = np.random.default_rng(seed)
        z = rng.standard_normal((B, D, H, W, K)) * 2.0
        lab = rng.integers(-1 if stray else 0, K + (1 if stray else 0), size=(B, D, H, W, 1)).astype(np.int32)
        wts = list(rng.uniform(0.1, 1.0, K))
        Z = O.Var(z)
        loss, sm = O.loss_head(Z, lab, loss_name, wts, alpha)
        O.backward(loss, 1.0)
        tz = g(z, dev).requires_grad_(True)
        tl, _, tsm, tpred = ops.softmax_loss(tz, g(lab, dev, torch.int32), loss_name, wts, alpha, want_softmax=True, want_pred=True)
        tag = "%s B%d K%d [%d,%d,%d] stray%d" % (loss_name, B, K, D, H, W, stray)
        check_close(tag + " loss", tl, loss.v, 2e-6, atol=2e-6)
        check_close(tag + " softmax", tsm, sm.v, 2e-6)
        assert (tpred.cpu().numpy() == O.argmax_pred(z)).all()
        tl.backward()
        check_close(tag + " dlogits", tz.grad, Z.g, 1e-5, atol=1e-9)

    run()


def test_dice_coe_known_answers(dev):
    """SURVEY 8(c) known answers: dice(t,t)=1 exactly for one-hot t; dice(p,0)=s/(sum p+s)."""
    from vnet_tensorflow_amd import model
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 3, size=(2, 4, 5, 6))
    t = g(O.one_hot(lab, 3), dev)
    assert float(model.dice_coe(t, t, 'sorensen')) == 1.0
    p = torch.softmax(g(rng.standard_normal((2, 4, 5, 6, 3)), dev), -1)
    d0 = float(model.dice_coe(p, torch.zeros_like(p), 'sorensen'))
    ref = np.mean(1e-5 / (p.cpu().numpy().astype(np.float64).sum((1, 2, 3)) + 1e-5))
    assert abs(d0 - ref) < 1e-9
    for kind in ("sorensen", "jaccard"):
        for w in ([], [0.2, 0.5, 1.0]):
            P = O.Var(p.cpu().numpy().astype(np.float64))
            d = O.dice_coe(P, t.cpu().numpy(), kind, weights=w)
            O.backward(d)
            tp = p.clone().requires_grad_(True)
            td = model.dice_coe(tp, t, kind, weights=w)
            check_close("dice_coe %s %s" % (kind, w), td, d.v, 2e-6)
            td.backward()
            check_close("dice_coe grad %s %s" % (kind, w), tp.grad, P.g, 1e-5)


def test_activation_standalone(dev):
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 4, 5, 16))
    x[0, 0, 0, 0, :4] = 0.0      # exact zeros: TF tie rule gives gradient 0
    a = rng.uniform(0.05, 0.3, 16)
    X, A_ = O.Var(x), O.Var(a)
    y = O.prelu(X, A_)
    dy = rng.standard_normal(x.shape)
    O.backward(y, dy)
    tx, ta = g(x, dev).requires_grad_(True), g(a, dev).requires_grad_(True)
    ty = ops.activation(tx, "prelu", ta)
    check_close("prelu fwd", ty, y.v, 1e-6)
    ty.backward(g(dy, dev))
    check_close("prelu dx", tx.grad, X.g, 1e-6)
    check_close("prelu dalpha", ta.grad, A_.g, 2e-6)


def test_hard_metrics(dev):
    """reference model.py:588-626: accuracy / tp,tn,fp,fn / sensitivity / specificity / hard dice vs a numpy count."""
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(2)
    K = 3
    lab = rng.integers(0, K, size=(2, 9, 10, 11))
    pred = np.where(rng.uniform(size=lab.shape) < 0.8, lab, rng.integers(0, K, size=lab.shape))
    m = ops.hard_metrics(g(pred, dev, torch.int64), g(lab, dev, torch.int32), K)
    assert abs(m["accuracy"] - (pred == lab).mean()) < 1e-12
    ref = O.hard_dice(pred, lab, K)
    for c in range(K):
        p, t = pred == c, lab == c
        assert m[c]["tp"] == (p & t).sum() and m[c]["fp"] == (p & ~t).sum() and m[c]["fn"] == (~p & t).sum() and m[c]["tn"] == (~p & ~t).sum()
        assert abs(m[c]["dice"] - ref[c]) < 1e-12
        assert abs(m[c]["sensitivity"] - (p & t).sum() / t.sum()) < 1e-12 and abs(m[c]["specificity"] - (~p & ~t).sum() / (~t).sum()) < 1e-12


def test_dropout(dev):
    from vnet_tensorflow_amd import ops
    x = torch.ones(4, 8, 8, 8, 16, device=dev, requires_grad=True)
    y = ops.dropout(x, 0.25)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.75) < 0.01
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1.0 / 0.75))
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0)
    assert ops.dropout(x, 0.0) is x


def test_optimisers(dev):
    from vnet_tensorflow_amd import optim
    rng = np.random.default_rng(5)
    shapes = [(5, 5, 5, 4, 8), (8,), (3,), (2, 2, 2, 8, 16)]
    vals = {("v%d" % i): rng.standard_normal(s) for i, s in enumerate(shapes)}
    for name in ("Adam", "SGD", "Momentum", "NesterovMomentum"):
        params = [(k, torch.nn.Parameter(g(v, dev))) for k, v in vals.items()]
        flat = optim.FlatParams(params)
        opt = optim.make_optimizer(name, flat, 0.9)
        ref = {k: v.copy() for k, v in vals.items()}
        ro = O.TFAdam() if name == "Adam" else O.TFMomentum(0.9, name == "NesterovMomentum") if "Momentum" in name else None
        for step in range(3):
            grads = {k: rng.standard_normal(v.shape) for k, v in vals.items()}
            lr = optim.exponential_decay(1e-2, step, 100, 0.99)
            flat.zero_grad()
            for k, p in params:
                p.grad.copy_(g(grads[k], dev))
            opt.apply(lr)
            ref = ro.step(ref, grads, lr) if ro else O.sgd_step(ref, grads, lr)
        for k, p in params:
            check_close("%s %s" % (name, k), p, ref[k], 2e-6)

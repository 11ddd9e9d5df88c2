"""CPU: pins the oracle (parity is unpinned by the reference -- it has no tests and TF 1.15 cannot run
here): two independent restatements agree in float64, hand-derivable known answers,
central-difference gradient checks, and the committed golden vectors reproduce."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as T
from oracle import vnet_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("cin,K,variant,loss,wts", [
    (1, 2, "networks", "sorensen", ()), (2, 3, "networks", "weighted_sorensen", (0.1, 0.5, 1.0)),
    (1, 2, "legacy", "sorensen", ()), (3, 4, "legacy", "jaccard", ()),
    (2, 3, "networks", "mixed_weighted_jaccard", (0.1, 0.5, 1.0)), (2, 3, "networks", "xent", ()),
])
def test_numpy_oracle_equals_torch_wiring(cin, K, variant, loss, wts):
    ps = O.ParamStore(rng=np.random.default_rng(7), perturb=0.2)
    net = O.VNetOracle(K, 0.0, 4, 2, (1, 2), 2, "prelu", variant, ps)
    x, lab = O.synthetic_batch(2, 8, cin, K)
    r = O.run_step(x.astype(np.float64), lab, net, loss, wts)
    params = {k: torch.tensor(v.v, dtype=torch.float64, requires_grad=True) for k, v in ps.vars.items()}
    tn = T.TorchVNet(K, 4, 2, (1, 2), 2, "prelu", variant, params)
    lg = tn.forward(torch.tensor(x, dtype=torch.float64))
    ls, _ = T.loss_head(lg, torch.tensor(lab), loss, wts)
    ls.backward()
    assert np.abs(lg.detach().numpy() - r["logits"]).max() < 1e-10
    assert abs(float(ls.detach()) - r["loss"]) < 1e-12
    assert set(params) == set(r["grads"])
    for k, p in params.items():
        gt = p.grad.numpy() if p.grad is not None else np.zeros_like(r["grads"][k])
        assert np.abs(gt - r["grads"][k]).max() <= 1e-9 * max(1.0, np.abs(gt).max()), k


def test_variable_catalogue_matches_survey():
    """241 variables, 43,940,486 trainable, 6,532 non-trainable, 38 BN layers, 29 alphas."""
    ps = O.ParamStore(rng=np.random.default_rng(0))
    net = O.VNetOracle(2, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", ps)
    net.GetNetwork(np.zeros((1, 16, 16, 16, 1)))
    assert sum(v.v.size for v in ps.vars.values()) == 43940486
    assert sum(v.size for v in ps.state.values()) == 6532
    assert len(ps.vars) + len(ps.state) == 241
    assert sum(k.endswith("/gamma") for k in ps.vars) == 38
    assert sum(k.endswith("/alpha") for k in ps.vars) == 29
    assert ps.vars["vnet/decoder/level_1/up_convolution/weights"].v.shape == (2, 2, 2, 16, 32)
    assert ps.vars["vnet/decoder/level_1/conv_1/batch_normalization_2/gamma"].v.shape == (16,)


def test_conv_known_answers():
    x = np.random.default_rng(0).standard_normal((1, 6, 7, 8, 3))
    w = np.zeros((5, 5, 5, 3, 3)); w[2, 2, 2] = np.eye(3)
    assert np.allclose(O.conv_nd_fwd(x, w, 1), x)                       # delta kernel
    ones = O.conv_nd_fwd(np.ones((1, 8, 8, 8, 2)), np.ones((5, 5, 5, 2, 1)), 1)[0, ..., 0]
    assert ones[4, 4, 4] == 125 * 2 and ones[0, 0, 0] == 27 * 2 and ones[0, 0, 4] == 45 * 2 and ones[0, 4, 4] == 75 * 2
    # SAME on odd sizes, k=2 s=2: out=ceil(in/2), padding on the high side only
    y = O.conv_nd_fwd(np.ones((1, 5, 5, 5, 1)), np.ones((2, 2, 2, 1, 1)), 2)[0, ..., 0]
    assert y.shape == (3, 3, 3) and y[0, 0, 0] == 8 and y[2, 2, 2] == 1 and y[2, 0, 0] == 4
    # down conv touches each voxel exactly once
    xx = np.random.default_rng(1).standard_normal((1, 4, 4, 4, 2))
    assert np.isclose(O.conv_nd_fwd(xx, np.ones((2, 2, 2, 2, 1)), 2).sum(), xx.sum())
    # up conv of a single voxel = the 2^3 filter block
    wt = np.random.default_rng(2).standard_normal((2, 2, 2, 3, 2))
    xi = np.zeros((1, 2, 2, 2, 2)); xi[0, 1, 0, 1, 1] = 1.0
    up = O.conv_nd_transpose_fwd(xi, wt, (4, 4, 4), 2)
    assert np.allclose(up[0, 2:4, 0:2, 2:4, :], wt[..., 1]) and np.isclose(np.abs(up).sum(), np.abs(wt[..., 1]).sum())
    # adjointness <down(x), y> == <x, up(y)>
    wd = np.random.default_rng(3).standard_normal((2, 2, 2, 3, 5))
    xa, ya = np.random.default_rng(4).standard_normal((1, 4, 6, 8, 3)), np.random.default_rng(5).standard_normal((1, 2, 3, 4, 5))
    assert np.isclose((O.conv_nd_fwd(xa, wd, 2) * ya).sum(), (xa * O.conv_nd_transpose_fwd(ya, wd, (4, 6, 8), 2)).sum())


def test_bn_prelu_dice_known_answers():
    rng = np.random.default_rng(0)
    x = O.Var(rng.standard_normal((2, 3, 4, 5, 6)) * 3 + 2)
    gmm, bt = rng.uniform(0.5, 2, 6), rng.standard_normal(6)
    y = O.batch_norm_train(x, O.Var(gmm), O.Var(bt)).v
    var = x.v.reshape(-1, 6).var(0)
    assert np.allclose(y.reshape(-1, 6).mean(0), bt) and np.allclose(y.reshape(-1, 6).var(0), gmm ** 2 * var / (var + 1e-3))
    # prelu tie: gradient 0 at exactly 0
    xv = O.Var(np.array([[-1.0, 0.0, 2.0]])); a = O.Var(np.array([0.1, 0.1, 0.1]))
    p = O.prelu(xv, a); O.backward(p, np.ones((1, 3)))
    assert np.allclose(p.v, [[-0.1, 0, 2]]) and np.allclose(xv.g, [[0.1, 0.0, 1.0]]) and np.allclose(a.g, [-1, 0, 0])
    lab = rng.integers(0, 3, (2, 4, 4, 4))
    t = O.one_hot(lab, 3)
    assert float(O.dice_coe(O.Var(t), t, 'sorensen').v) == 1.0
    pu = np.full(t.shape, 1 / 3.0)
    d = O.dice_coe(O.Var(pu), t, 'sorensen').v
    n_c = t.sum((1, 2, 3)); N = 64
    assert np.isclose(d, ((2 * n_c / 3 + 1e-5) / (N / 3 + n_c + 1e-5)).mean())
    assert np.isclose(O.dice_coe(O.Var(pu), np.zeros_like(t), 'sorensen').v, 1e-5 / (N / 3 + 1e-5))
    assert O.one_hot(np.array([5, -1]), 3).sum() == 0               # out-of-range labels -> zero rows
    assert (O.argmax_pred(np.array([[1.0, 1.0, 0.5]])) == [0]).all()  # ties -> lowest index


def test_gradient_check_central_differences():
    ps = O.ParamStore(rng=np.random.default_rng(3), perturb=0.2)
    net = O.VNetOracle(2, 0.0, 2, 1, (2,), 1, "prelu", "networks", ps)
    x, lab = O.synthetic_batch(1, 4, 2, 2, seed=5)
    x = x.astype(np.float64) / 50.0
    r = O.run_step(x, lab, net, "mixed_sorensen", (), 0.5)
    rng = np.random.default_rng(0)
    for name in ["vnet/input_layer/weights", "vnet/encoder/level_1/conv_2/weights", "vnet/encoder/level_1/conv_1/alpha",
                 "vnet/decoder/level_1/up_convolution/weights", "vnet/decoder/level_1/conv_2/batch_normalization_1/gamma",
                 "vnet/output_layer/weights", "vnet/bottom_level/conv_1/biases"]:
        v = ps.vars[name]
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in v.v.shape)
            old = v.v[idx]
            v.v[idx] = old + 1e-5; lp = O.run_step(x, lab, net, "mixed_sorensen", (), 0.5, want_grads=False)["loss"]
            v.v[idx] = old - 1e-5; lm = O.run_step(x, lab, net, "mixed_sorensen", (), 0.5, want_grads=False)["loss"]
            v.v[idx] = old
            fd = (lp - lm) / 2e-5
            assert abs(fd - r["grads"][name][idx]) < 1e-6 + 1e-4 * abs(fd), (name, idx, fd, r["grads"][name][idx])


def test_tf_adam_form_and_lr_schedule():
    p, gr = {"a": np.array([1.0, -2.0])}, {"a": np.array([0.5, -0.25])}
    out = O.TFAdam().step(dict(p), gr, 0.1)
    m, v = 0.1 * gr["a"], 0.001 * gr["a"] ** 2
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(out["a"], p["a"] - lr_t * m / (np.sqrt(v) + 1e-8))
    assert np.isclose(O.exponential_decay(1e-2, 50, 100, 0.99), 1e-2 * 0.99 ** 0.5)


def test_golden_vectors_reproduce():
    """The committed fixtures are what the oracle produces today (guards oracle and fixtures together)."""
    from tests.golden.make_golden import SMALL
    name = "small_networks_c1k2"
    variant, cin, K, P, B, C0, levels, ncv, nb, loss, wts = SMALL[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    values = {k[6:]: z[k].astype(np.float64) for k in z.files if k.startswith("param:")}
    ps = O.ParamStore(values=values)
    net = O.VNetOracle(K, 0.0, C0, levels, ncv, nb, "prelu", variant, ps)
    r = O.run_step(z["images"].astype(np.float64), z["labels"], net, loss, wts, 0.7)
    assert abs(r["loss"] - float(z["loss"])) < 1e-7
    assert np.abs(r["logits"] - z["logits"]).max() < 1e-4
    for k in ps.vars:
        assert np.abs(r["grads"][k] - z["grad:" + k]).max() <= 1e-5 * max(1.0, np.abs(z["grad:" + k]).max()), k


def test_bf16_operand_rounding_mode():
    """BASELINE config C5 arithmetic in the oracle: round-to-nearest-even to bfloat16 known answers, agreement
    with torch's own bf16 cast, and the committed bf16 fixture reproduces (and differs from the exact one by a
    bf16-sized amount, not more)."""
    import torch
    from tests.golden.make_golden import SMALL_BF16
    assert O.round_bf16(1.0) == 1.0 and O.round_bf16(-2.5) == -2.5
    assert O.round_bf16(1.0 + 2.0 ** -8) == 1.0                     # tie -> even mantissa
    assert O.round_bf16(1.0 + 3 * 2.0 ** -8) == 1.0 + 2.0 ** -6     # tie -> even (up)
    assert O.round_bf16(1.0 + 2.0 ** -8 + 2.0 ** -20) == 1.0 + 2.0 ** -7
    a = np.random.default_rng(5).standard_normal(20000) * np.exp(np.random.default_rng(6).uniform(-20, 20, 20000))
    assert np.array_equal(O.round_bf16(a), torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).double().numpy())
    name = "small_networks_c4k5_bf16"
    variant, cin, K, P, B, C0, levels, ncv, nb, loss, wts = SMALL_BF16[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ps = O.ParamStore(values={k[6:]: z[k].astype(np.float64) for k in z.files if k.startswith("param:")})
    net = O.VNetOracle(K, 0.0, C0, levels, ncv, nb, "prelu", variant, ps)
    O.CONV5_OPERAND_ROUNDING = "bf16"
    try:
        r = O.run_step(z["images"].astype(np.float64), z["labels"], net, loss, wts, 0.7)
    finally:
        O.CONV5_OPERAND_ROUNDING = None
    assert abs(r["loss"] - float(z["loss"])) < 1e-7
    assert np.abs(r["logits"] - z["logits"]).max() < 1e-4
    exact = np.load(os.path.join(GOLD, "small_networks_c4k5.npz"))
    d = np.linalg.norm(exact["logits"] - z["logits"]) / np.linalg.norm(exact["logits"])
    assert 1e-3 < d < 5e-2, d


@pytest.mark.parametrize("cin,K,variant", [(1, 2, "networks"), (4, 5, "networks"), (2, 3, "legacy")])
def test_bf16_storage_mode_two_restatements_agree(cin, K, variant):
    """Round 3, config C5 as SURVEY 8(d) words it: bf16 STORAGE of every activation and activation gradient, bf16 operands in
    every spatial convolution.  The NumPy tape (ACT_STORAGE, `store()` marks) and the independent torch-autograd wiring
    (torch_ref.STORAGE: torch's own bf16 conversion in a custom Function, straight-through filter rounding) round at the same
    places: logits equal to float64 round-off, stored tensors are bf16-representable, and the gradients agree as closely as one
    flipped rounding allows (two float64 sums that differ in their last bit can round a stored value to different bf16
    neighbours; batch-norm backward amplifies that one ulp) -- i.e. a misplaced or missing rounding point, which moves
    every gradient by ~1e-2, cannot hide."""
    ps = O.ParamStore(rng=np.random.default_rng(11), perturb=0.2)
    net = O.VNetOracle(K, 0.0, 8, 2, (1, 2), 2, "prelu", variant, ps)
    x, lab = O.synthetic_batch(2, 8, cin, K)
    O.ACT_STORAGE = "bf16"
    T.STORAGE = "bf16"
    try:
        r = O.run_step(x.astype(np.float64), lab, net, "sorensen")
        params = {k: torch.tensor(v.v, dtype=torch.float64, requires_grad=True) for k, v in ps.vars.items()}
        tn = T.TorchVNet(K, 8, 2, (1, 2), 2, "prelu", variant, params)
        lg = tn.forward(torch.tensor(x, dtype=torch.float64))
        ls, _ = T.loss_head(lg, torch.tensor(lab), "sorensen")
        ls.backward()
        # a stored tensor is bf16-representable: conv output of the first encoder block, as the tape holds it
        y = O.store(O.Var(np.random.default_rng(0).standard_normal(1000)))
        assert np.array_equal(y.v, O.round_bf16(y.v))
    finally:
        O.ACT_STORAGE = None
        T.STORAGE = None
    assert np.abs(lg.detach().numpy() - r["logits"]).max() < 1e-9
    assert abs(float(ls.detach()) - r["loss"]) < 1e-11
    exact = O.run_step(x.astype(np.float64), lab, O.VNetOracle(K, 0.0, 8, 2, (1, 2), 2, "prelu", variant,
                                                               O.ParamStore(values={k: v.v for k, v in ps.vars.items()})), "sorensen")
    d = np.linalg.norm(exact["logits"] - r["logits"]) / np.linalg.norm(exact["logits"])
    assert 1e-4 < d < 1e-1, d                         # the mode does round, by a bf16-sized amount
    worst = 0.0
    for k, p in params.items():
        gt = p.grad.numpy() if p.grad is not None else np.zeros_like(r["grads"][k])
        n = np.linalg.norm(r["grads"][k])
        if n > 1e-12:
            worst = max(worst, float(np.linalg.norm(gt - r["grads"][k]) / n))
    assert worst < 1e-8, worst       # (measured 4e-14 .. 7e-12 on these seeded cases: no rounding flipped)


# ---- the reference's own graph-building code, executed once in the build container (tests/golden/make_ref_wiring.py) ------------
_WIRING = sorted(f[len("ref_wiring_"):-4] for f in os.listdir(os.path.join(os.path.dirname(__file__), "golden"))
                 if f.startswith("ref_wiring_") and f.endswith(".npz"))


@pytest.mark.parametrize("case", _WIRING)
def test_reference_wiring_matches_oracle_and_product_names(case):
    """Fixtures made by RUNNING networks.VNet.GetNetwork (networks.py:246-365) / VNet.VNet.network_fn (VNet.py:26-155) / layers2
    (layers2.py:59-99) / model.dice_coe (model.py:26-85) from /root/reference against a NumPy-eager stand-in for tf.* (the
    stand-in's arithmetic and TF naming rules are this repo's: this pins WIRING, creation ORDER and variable NAMES -- the checkpoint
    contract -- not TensorFlow's numerics; parity stays unpinned by the reference).  Checked here: the oracle creates the same
    trainables in the same order with the same shapes and the same moving-statistics set, its logits, its moving-average updates
    and its three dice_coe forms agree to 1e-10; the product's variable store (vnet_tensorflow_amd/_scope.py via networks / VNet)
    creates the same names, shapes and order."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_wiring_%s.npz" % case))
    variant, cin, K, C, levels, convs, bottom, act = [str(v) for v in z["config"]]
    cin, K, C, levels, bottom = int(cin), int(K), int(C), int(levels), int(bottom)
    convs = tuple(int(v) for v in convs.split(","))
    names = [str(n) for n in z["names"]]
    trainable = [bool(t) for t in z["trainable"]]
    shapes = [tuple(int(v) for v in str(s).split(",")) if str(s) else () for s in z["shapes"]]
    values = {n: z["v:" + n].astype(np.float64) for n in names}
    x = z["x"].astype(np.float64)

    ps = O.ParamStore(rng=np.random.default_rng(0), values=values)
    net = O.VNetOracle(K, 0.0, C, levels, convs, bottom, act, variant, ps)
    for n in names:                                                     # injected moving statistics
        if not trainable[names.index(n)]:
            ps.state[n] = values[n].copy()
    logits = net.GetNetwork(x)
    want_tr = [(n, s) for n, s, t in zip(names, shapes, trainable) if t]
    assert [(n, tuple(ps.vars[n].v.shape)) for n in ps.order] == want_tr
    assert set(ps.state) == {n for n, t in zip(names, trainable) if not t}
    # tf.layers.BatchNormalization.build order: gamma, beta, moving_mean, moving_variance, consecutively
    for i, n in enumerate(names):
        if n.endswith("/gamma"):
            base = n[:-len("gamma")]
            assert names[i + 1:i + 4] == [base + "beta", base + "moving_mean", base + "moving_variance"]
    got = logits.v if hasattr(logits, "v") else logits
    assert np.abs(got - z["logits"]).max() < 1e-10 * max(1.0, np.abs(z["logits"]).max())
    for n in ps.state:                                                  # the update ops of one training step (incl. the dead batch-norms)
        assert np.abs(ps.state[n] - z["u:" + n]).max() < 1e-10, n
    # model.dice_coe as the loss switch calls it (model.py:503-504, 70-75)
    zc = z["logits"] - z["logits"].max(-1, keepdims=True)
    sm = np.exp(zc) / np.exp(zc).sum(-1, keepdims=True)
    oh = np.eye(K)[z["labels"].astype(np.int64)]
    ax = (1, 2, 3)
    assert abs(float(O.dice_coe(O.const(sm), oh, 'sorensen', ax).v) - float(z["dice_sorensen"])) < 1e-12
    assert abs(float(O.dice_coe(O.const(sm), oh, 'jaccard', ax).v) - float(z["dice_jaccard"])) < 1e-12
    assert abs(float(O.dice_coe(O.const(sm), oh, 'sorensen', ax, weights=list(z["dice_weights"])).v) - float(z["dice_weighted_sorensen"])) < 1e-12

    # the product's mirrored modules create the same variables (names, shapes, creation order)
    from vnet_tensorflow_amd import networks, VNet
    shape = tuple(x.shape)
    if variant == "networks":
        pnet = networks.VNet(K, 0.0, C, levels, convs, bottom, True, act, device="cpu").build(shape)
    else:
        pnet = VNet.VNet(K, 1.0, C, levels, convs, bottom, True, act, device="cpu").build(shape)
    assert [(n, tuple(p.shape)) for n, p in pnet.named_parameters()] == want_tr
    assert set(pnet.variables.buffers) == {n for n, t in zip(names, trainable) if not t}

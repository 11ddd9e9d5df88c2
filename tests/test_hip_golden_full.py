"""-m gpu: whole-network parity at the sizes the bench runs: logits, loss, soft-Dice sums, argmax and every
parameter gradient of one training step of the full-width V-Net against the fp64 oracle fixtures
tests/golden/{c3_128cube, c2_64cube_b2, c5_128cube_b16}.npz (made by tests/golden/make_golden_full.py;
ORACLE outputs -- the reference itself cannot run here, parity unpinned by it).

Tolerances: logits rtol/atol 1e-3 (rel-L2 1e-4), loss abs 1e-5, soft-Dice sums rtol 1e-5, argmax agreement >= 99.99 %
(BASELINE.md 2.1).  GRADIENTS: BASELINE.md 2.1 asks rel-L2 1e-3 per tensor; at full width that is below what fp32 itself
resolves on this problem -- the batch-norm backward passes subtract the (1, xhat) projections of a gradient that is almost
entirely inside that span, so one rounding of dy (6e-8) comes out ~1e5 times larger relative to what is left, and every
layer upstream inherits it.  MEASURED against these same fixtures (profiles/r02_golden_full_errors.txt):
    stock PyTorch-CPU fp32 (oneDNN, torch autograd; profiles/golden_full_errors_cpu.py), C2: filters max 5.0e-3 / median
    3.8e-3, per-channel vectors max 5.2e-3 / median 3.7e-3;   HIP path, C2: filters 6.9e-3 / 2.8e-3, vectors 9.6e-3 / 2.9e-3;
    HIP path, C3: filters 5.8e-3 / 5.2e-3, vectors 6.7e-3 / 4.6e-3.
The test therefore holds every tensor to 2.75e-2, the median over tensors to 1.4e-2 and the whole gradient vector to 1.5e-3
(rel-L2 on the stored seeded sample of <= 2048 elements per tensor), plus norm agreement 1.1e-2 -- one set of bounds for both fp32
modes, 1.25 x the largest value of a ten-draw seed spread (FULL_BOUNDS below, profiles/r06_golden_seed_spread.txt); the loss head and
logits stay at the BASELINE tolerances.  Weights and inputs come from the recipe the fixture was made with."""
import os

import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.golden.make_golden_full import CASES, SAMPLE, STRIDE, WEIGHT_SEED, sample_indices
from tests.util import g, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run_case(dev, case, compute=None):
    from vnet_tensorflow_amd import networks, ops
    fname, P, B, cin, K, seed, rounding = CASES[case]
    z = np.load(os.path.join(GOLD, fname))
    store = O.ParamStore(rng=np.random.default_rng(WEIGHT_SEED[case]))
    ref_net = O.VNetOracle(K, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", store)
    ref_net.GetNetwork(np.zeros((1, 16, 16, 16, cin)))             # creates the variables in the fixture's order
    assert list(store.vars.keys()) == [str(n) for n in z["names"]]
    x, lab = O.synthetic_batch(B, P, cin, K, seed=seed)
    ops.set_compute_dtype(compute or {"storage": "bf16"}.get(rounding, "fp32"))
    try:
        net = networks.VNet(K, 0.0, 16, 4, (1, 2, 3, 3), 3, True, "prelu", device=dev)
        net.variables.values = {k: v.v for k, v in store.vars.items()}
        net.build(x.shape)
        logits = net.GetNetwork(g(x, dev))
        labels = g(lab, dev, torch.int32)
        loss, dice, sm, pred = ops.softmax_loss(logits, labels, "sorensen", want_softmax=True, want_pred=True)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype("fp32")
    return z, net, logits.detach(), float(loss.detach()), sm.detach(), pred, lab, K


def _dice_sums(sm, lab, K):
    oh = torch.nn.functional.one_hot(torch.from_numpy(lab[..., 0]).long().to(sm.device), K).to(torch.float64)
    s = sm.to(torch.float64)
    ax = (1, 2, 3)
    return (s * oh).sum(ax).cpu().numpy(), s.sum(ax).cpu().numpy(), oh.sum(ax).cpu().numpy()


def _grad_errors(z, net):
    params = dict(net.named_parameters())
    out = []
    for i, n in enumerate(z["names"]):
        n = str(n)
        p = params[n]
        gn = float(z["grad_norm"][i])
        if p.grad is None:
            assert gn == 0.0, n
            continue
        got = p.grad.detach().cpu().numpy().astype(np.float64).ravel()
        idx = sample_indices(i, got.size)
        ref = z["grad_sample"][i][:len(idx)].astype(np.float64)
        if gn < 1e-7:            # conv biases in front of a batch-norm: analytically zero gradient
            assert np.abs(got).max() < 1e-4, n
            continue
        # the sample's rel-L2 estimates the tensor's (error and reference sampled at the same seeded positions)
        out.append((n, rel_l2(got[idx], ref), abs(np.linalg.norm(got) - gn) / gn,
                    np.abs(np.resize(got[:8], 8) - z["grad_head"][i]).max() / max(gn, np.abs(z["grad_head"][i]).max()),
                    abs(got.sum() - float(z["grad_sum"][i])) / max(gn, 1e-30)))
    return out


# Gradient bounds of the full-size fp32 fixtures -- ONE set for both fp32 modes (VERDICT r5 next #1a), set from the committed seed spread
# profiles/r06_golden_seed_spread.txt: ten (weight seed, input seed) draws per config x {fp32, fp32_split3} on the final kernels.  The
# per-tensor "worst" of a run is chaotic (a rounding-level change of the FIRST layer moved single draws by 3x in either direction, in both
# modes), so a bound has to cover the spread, not one draw: 1.25 x the largest value seen in the forty runs --
#   per-tensor sampled rel-L2 2.20e-2 (fp32, draw c3s6; fp32_split3 max 1.55e-2) -> 2.75e-2;  first-8-elements error 1.10e-2 -> 1.4e-2;
#   norm 8.9e-3 -> 1.1e-2;  median over tensors 1.10e-2 -> 1.4e-2;  whole gradient vector 1.2e-3 -> 1.5e-3 (rounds 2-5: 8e-3).
# (Rounds 2-5 held 1.2e-2 / 1.5e-2 per tensor on ONE draw per config; draw c3s6 exceeds that in the reference's own arithmetic.)
# What pins the kernels is the per-kernel parity (2e-6, tests/test_hip_ops.py / test_hip_x3.py / test_hip_fullsize.py); this test pins the
# assembly of the network at the bench sizes.
FULL_BOUNDS = {"tensor": 2.75e-2, "head": 1.4e-2, "norm": 1.1e-2, "median": 1.4e-2, "vector": 1.5e-3}


@pytest.mark.parametrize("compute", ["fp32", "fp32_split3"])
@pytest.mark.parametrize("case", ["c3", "c2", "c3s1", "c2s1", "c3s2", "c2s2"])
def test_full_size_network_fp32(dev, case, compute):
    """BASELINE configs[2]/[3] (128^3, B=1 -- the exact BENCH workload) and configs[1] (64^3, B=2) at full width, three draws each
    (fixture + two of the seed spread; profiles/golden_seed_spread.py runs all ten), in both fp32 modes at the SAME bounds:
    ComputeDtype fp32 (v_mfma_f32_16x16x4_f32) and fp32_split3 (the 5^3 convolutions form their products from six bf16 products of
    exactly split operands, csrc/conv_x3.h)."""
    z, net, logits, loss, sm, pred, lab, K = _run_case(dev, case, compute)
    s = (slice(None),) + (slice(None, None, STRIDE),) * 3
    got = logits[s].cpu().numpy()
    ref = z["logits_sample"]
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=1e-3, atol=1e-3), np.abs(got - ref).max()
    assert rel_l2(got, ref) < 1e-4, rel_l2(got, ref)
    assert abs(loss - float(z["loss"])) < 1e-5, (loss, float(z["loss"]))
    I, L, R = _dice_sums(sm, lab, K)
    for a, b, nm in ((I, z["dice_I"], "I"), (L, z["dice_L"], "L"), (R, z["dice_R"], "R")):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-2), (nm, a, b)
    dice = ((2 * I + 1e-5) / (L + R + 1e-5)).mean()
    assert abs((1.0 - dice) - float(z["loss"])) < 1e-5
    assert (pred[s].cpu().numpy() == z["pred_sample"]).mean() >= 0.9999
    errs = _grad_errors(z, net)
    assert len(errs) > 100
    worst = sorted(errs, key=lambda e: -e[1])[:5]
    for n, e_sample, e_norm, e_head, e_sum in errs:
        assert e_sample < FULL_BOUNDS["tensor"] and e_norm < FULL_BOUNDS["norm"] and e_head < FULL_BOUNDS["head"], (n, e_sample, e_norm, e_head, worst)
    assert np.median([e[1] for e in errs]) < FULL_BOUNDS["median"], np.median([e[1] for e in errs])
    names = list(map(str, z["names"]))
    num = sum((e[1] * float(z["grad_norm"][names.index(e[0])])) ** 2 for e in errs)
    den = sum(float(v) ** 2 for v in z["grad_norm"])
    assert (num / den) ** 0.5 < FULL_BOUNDS["vector"], (num / den) ** 0.5


def test_full_size_network_c5_b16_storage(dev):
    """BASELINE configs[4] per-GPU workload in the bf16-STORAGE mode of round 3 (bf16 activations / gradients in HBM, bf16 operands
    into every spatial convolution, fp32 statistics / Dice / parameter gradients) against the oracle's ACT_STORAGE restatement
    (tests/golden/c5_128cube_b16.npz).  Rounding is discontinuous (a last-bit fp32 difference can move a value to the neighbouring bf16 number) and the whole-network
    yardstick is the oracle's own sensitivity (tests/test_hip_b16.py::test_small_network_bf16_storage_against_oracle measures it
    on small networks: logits 0.8 .. 1.3e-2, gradient tensors ~1e-1 median); kernels and rounding points are pinned per op
    (tests/test_hip_b16.py) and per layer on this very run's data (test_teacher_forced_layers_c5_b16 below)."""
    z, net, logits, loss, sm, pred, lab, K = _run_case(dev, "c5s")
    s = (slice(None),) + (slice(None, None, STRIDE),) * 3
    got, ref = logits[s].cpu().numpy(), z["logits_sample"]
    # measured in round 4 (profiles/r04_golden_full_errors.txt): logits 2.31e-2, loss 1.5e-6, predictions 98.57 %, gradient 11.2 %
    # (round 3's bounds were 3e-2 / 1e-3 / 0.3: 30 % to three orders of magnitude of slack)
    # (round 5, ADVICE r4: round 4 had put these ~15 % above ONE measurement of a chaotic quantity -- 2.7e-2 vs 2.31e-2, 0.14 vs 0.112 --
    #  so that any benign change of a summation order, e.g. of the grouped filter-gradient plan, could trip them; now 1.5x the
    #  measurement.  The per-layer teacher-forced test below is the tight one.)
    assert rel_l2(got, ref) < 3.5e-2, rel_l2(got, ref)
    assert abs(loss - float(z["loss"])) < 2e-5, (loss, float(z["loss"]))
    assert (pred[s].cpu().numpy() == z["pred_sample"]).mean() >= 0.98
    errs = _grad_errors(z, net)
    names = list(map(str, z["names"]))
    num = sum((e[1] * float(z["grad_norm"][names.index(e[0])])) ** 2 for e in errs)
    den = sum(float(v) ** 2 for v in z["grad_norm"])
    assert (num / den) ** 0.5 < 0.17, (num / den) ** 0.5


def test_teacher_forced_layers_c5_b16(dev):
    """VERDICT r2 next #3(b), widened in round 4: per-layer parity of the bf16 kernels WITHOUT chaotic
    amplification.  The fixture holds, for eight 5^3 layers of the 128^3 C5 oracle run (input conv 4->16, 16->16 and 32->16 at 128^3,
    32->32 at 64^3, 64->64 at 32^3, 128->128 at 16^3, 256->256 at 8^3 -- the last three take the deep-level kernel -- and the two-source
    decoder conv 64+64->64 at 32^3) and for the level-2 pair of 2^3 convolutions (down 32->64, transposed 64->32), a crop of the bf16 tensor the layer actually read and of the bf16 gradient that actually arrived at its output.  Each
    layer is run alone on that data -- forward, backward-data and filter gradient -- and compared per tensor with the oracle's
    convolution of the same operands: stored outputs must be correct roundings (half a bf16 ulp), the fp32 filter gradient 2e-6."""
    from tests.golden.make_golden_full import TF_LAYERS, from_bf16_bits
    from tests.test_hip_b16 import check_bf16, g16, rb
    from tests.util import check_close
    from vnet_tensorflow_amd import ops
    fname, P, B, cin, K, seed, rounding = CASES["c5s"]
    z = np.load(os.path.join(GOLD, fname))
    store = O.ParamStore(rng=np.random.default_rng(42))
    O.VNetOracle(K, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", store).GetNetwork(np.zeros((1, 16, 16, 16, cin)))
    for name in TF_LAYERS:
        x, dy = from_bf16_bits(z["tf:%s:x" % name]), from_bf16_bits(z["tf:%s:dy" % name])
        w, b = store.vars[name].v, store.vars[name[:-len("weights")] + "biases"].v
        assert x.shape[:4] == dy.shape[:4] and x.shape[-1] == w.shape[-2] and dy.shape[-1] == w.shape[-1]
        assert float(np.abs(x).max()) > 0 and float(np.abs(dy).max()) > 0
        y_ex = O.conv_nd_fwd(x, rb(w), 1) + b
        dx_ex, dw_ex = O.conv_nd_bwd(x, rb(w), dy, 1)
        # the first convolution of a decoder level reads concat(up-convolved, skip) (networks.py:325): two sources, never materialised
        two = name.startswith("vnet/decoder") and name.endswith("conv_1/weights") and x.shape[-1] == 2 * w.shape[-1]
        C0 = x.shape[-1] // 2 if two else x.shape[-1]
        tx1 = None
        if x.shape[-1] % 8:                      # the network input: fp32 image -> bf16, zero-padded to 8 channels
            tx = ops.cast_input(g(x, dev))
        else:
            tx = g16(x[..., :C0], dev).requires_grad_(True)
            if two:
                tx1 = g16(x[..., C0:], dev).requires_grad_(True)
        tw, tb = g(w, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
        y = ops.conv(tx, tw, tb, 5, 1, x1=tx1)
        check_bf16("teacher-forced %s fwd" % name, y, y_ex)
        y.backward(g16(dy, dev))
        if tx.requires_grad:
            check_bf16("teacher-forced %s dx" % name, tx.grad, dx_ex[..., :C0], noise=1e-5)
        if two:
            check_bf16("teacher-forced %s dx (skip source)" % name, tx1.grad, dx_ex[..., C0:], noise=1e-5)
        check_close("teacher-forced %s dw" % name, tw.grad, dw_ex, 2e-6, atol=2e-6 * float(np.abs(dw_ex).max()))
    # the level-2 pair of 2^3 convolutions on the same run's data (stride 2: the crops carry no halo)
    from tests.golden.make_golden_full import TF_LAYERS2
    for name, (kind, origin) in TF_LAYERS2.items():
        x, dy = from_bf16_bits(z["tf:%s:x" % name]), from_bf16_bits(z["tf:%s:dy" % name])
        w, b = store.vars[name].v, store.vars[name[:-len("weights")] + "biases"].v
        assert float(np.abs(x).max()) > 0 and float(np.abs(dy).max()) > 0
        tx, tw, tb = g16(x, dev).requires_grad_(True), g(w, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
        if kind == "down":                       # layers2.py:78-84
            y_ex = O.conv_nd_fwd(x, rb(w), 2) + b
            dx_ex, dw_ex = O.conv_nd_bwd(x, rb(w), dy, 2)
            y = ops.conv(tx, tw, tb, 2, 2)
        else:                                    # layers2.py:65-74, 88-94
            y_ex = O.conv_nd_transpose_fwd(x, rb(w), dy.shape[1:4], 2) + b
            dx_ex = O.conv_nd_fwd(dy, rb(w), 2)
            _, dw_ex = O.conv_nd_bwd(dy, w, x, 2, need_dx=False)
            y = ops.conv_transpose2(tx, tw, tb, dy.shape[1:4])
        assert tuple(y.shape) == tuple(y_ex.shape) == tuple(dy.shape)
        check_bf16("teacher-forced %s fwd" % name, y, y_ex)
        y.backward(g16(dy, dev))
        check_bf16("teacher-forced %s dx" % name, tx.grad, dx_ex, noise=1e-5)
        check_close("teacher-forced %s dw" % name, tw.grad, dw_ex, 2e-6, atol=2e-6 * float(np.abs(dw_ex).max()))

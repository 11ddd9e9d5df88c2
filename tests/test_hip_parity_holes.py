"""-m gpu: parity items VERDICT r1 listed as open (#8): dropout against the oracle with the kernel's own mask, the native
C++ inference driver against the ORACLE (not against the Python HIP path), tf.metrics.auc, streaming metrics."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.util import g, check_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vnet_tensorflow_amd", "vnet_infer")


@pytest.mark.parametrize("rate", [0.01, 0.25, 0.6])
def test_dropout_against_oracle_with_the_kernels_mask(dev, rate):
    """tf.nn.dropout(x, rate) (networks.py:321): the kernel emits its keep-mask; fed to oracle.dropout(x, rate, mask) the
    forward values and the input gradient must agree to fp32 round-off (1/(1-rate) scaling in fp32 vs fp64)."""
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(3)
    xs = rng.standard_normal((2, 6, 7, 5, 12))
    gs = rng.standard_normal(xs.shape)
    x = g(xs, dev).requires_grad_(True)
    y = ops.dropout(x, rate)
    mask = y.grad_fn.saved_tensors[0].cpu().numpy().astype(np.float64)       # the keep-mask the kernel wrote
    y.backward(g(gs, dev))
    assert set(np.unique(mask)) <= {0.0, 1.0} and abs(mask.mean() - (1.0 - rate)) < 0.05
    xv = O.Var(xs.astype(np.float32).astype(np.float64))
    ref = O.dropout(xv, rate, mask)
    xv.g = None
    O.backward(ref, seed=gs.astype(np.float32).astype(np.float64))
    check_close("dropout fwd", y, ref.v, 2e-7)
    check_close("dropout bwd", x.grad, xv.g, 2e-7)
    assert np.array_equal(y.detach().cpu().numpy() != 0, (mask != 0) & (xs.astype(np.float32) != 0))


def test_dropout_streams_differ_between_layers_and_steps(dev):
    """State-driven dropout (graph-replayable): the mask depends on (layer position in the pass, step number)."""
    from vnet_tensorflow_amd import ops
    x = torch.ones(1, 8, 8, 8, 16, device=dev)
    st = ops.step_state(dev)
    masks = []
    for step in (5, 6):
        ops.set_step_state(st, 0.0, 0.0, step)
        with ops.use_step_state(st):
            ops.begin_dropout_pass()
            masks.append([(ops.dropout(x, 0.5) != 0).cpu() for _ in range(2)])
    assert not torch.equal(masks[0][0], masks[0][1]) and not torch.equal(masks[0][0], masks[1][0])
    ops.set_step_state(st, 0.0, 0.0, 5)
    with ops.use_step_state(st):
        ops.begin_dropout_pass()
        assert torch.equal((ops.dropout(x, 0.5) != 0).cpu(), masks[0][0])        # reproducible for a given (layer, step)


@pytest.mark.parametrize("K,N", [(2, 20000), (5, 50000)])
def test_tf_metrics_auc(dev, K, N):
    """tf.metrics.auc defaults (200 thresholds, ROC, trapezoidal; model.py:607,613,624) from the GPU histogram against the
    oracle's direct per-threshold counts; predictions ON thresholds (k/199 as float32) exercise the strict `>`."""
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(K)
    lab = rng.integers(0, K, size=N).astype(np.int32)
    logits = rng.standard_normal((N, K)) + 1.5 * (lab[:, None] == np.arange(K))
    sm = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    sm = sm.astype(np.float32)
    th = ops.tf_auc_thresholds()
    sm[:400, 1] = th[rng.integers(0, 200, size=400)]            # exactly on a threshold
    sm[400:420, 1] = [0.0, 1.0] * 10
    smt, labt = g(sm, dev), g(lab, dev, torch.int32)
    for c in range(1, K):
        hist = ops.auc_histogram(smt, labt, K, c).cpu().numpy()
        assert hist.sum() == N and hist[0].sum() == (lab == c).sum()
        got, ref = ops.auc_from_hist(hist), O.tf_metrics_auc(lab == c, sm[:, c])
        assert abs(got - ref) < 1e-12, (c, got, ref)
        assert 0.5 < got <= 1.0
    # streaming: two updates == one update on the concatenation (TF's accumulating local variables)
    pred = torch.from_numpy(sm.argmax(-1)).to(dev)
    a = ops.StreamingMetrics(K).update(pred[:N // 2], labt[:N // 2], smt[:N // 2]).update(pred[N // 2:], labt[N // 2:], smt[N // 2:]).result()
    b = ops.StreamingMetrics(K).update(pred, labt, smt).result()
    for c in range(1, K):
        assert abs(a[c]["auc"] - b[c]["auc"]) < 1e-12 and a[c]["tp"] == b[c]["tp"] and a[c]["fn"] == b[c]["fn"]
        assert abs(b[c]["auc"] - O.tf_metrics_auc(lab == c, sm[:, c])) < 1e-12


def test_native_driver_against_the_oracle(tmp_path, dev):
    """csrc/vnet_infer.cpp (C ABI only) vs the ORACLE: per sliding-window patch the oracle's forward + softmax, accumulated
    and counted in NumPy exactly as model.py:919-937 does (sum of softmax over overlapping patches, argmax of the sums,
    probabilities = sums / counts) -- the duplicated last batch of model.py:903 included."""
    cin, K, levels, convs, bottom, batch = 2, 3, 2, [1, 2], 1, 2
    P, stride = (8, 8, 8), (4, 8, 6)
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    ps = O.ParamStore(rng=np.random.default_rng(4), perturb=0.2)
    net = O.VNetOracle(K, 0.0, 4, levels, tuple(convs), bottom, "prelu", "networks", ps)
    vol, _ = O.synthetic_batch(1, 14, cin, K, seed=12)
    vol = np.ascontiguousarray(vol[0][:, :12, :10]).astype(np.float32)            # 14 x 12 x 10
    dims = vol.shape[:3]
    # patch enumeration of model.py:866-903
    nums = [int(np.ceil((dims[a] - P[a]) / float(stride[a]))) + 1 for a in range(3)]
    idxs = []
    for i in range(nums[0]):
        for j in range(nums[1]):
            for k in range(nums[2]):
                o = [min(n * stride[a], dims[a] - P[a]) for a, n in enumerate((i, j, k))]
                idxs.append(o)
    batches = [idxs[b:b + batch] for b in range(0, len(idxs), batch)]
    batches.append(batches[-1])                                        # "for last batch" (model.py:903): appended once more
    acc = np.zeros(dims + (K,))
    cnt = np.zeros(dims)
    for bt in batches:
        xb = np.stack([vol[o[0]:o[0] + P[0], o[1]:o[1] + P[1], o[2]:o[2] + P[2]] for o in bt]).astype(np.float64)
        sm = O.softmax(net.GetNetwork(xb)).v                           # batch statistics of THIS batch (model.py:917)
        for o, s in zip(bt, sm):
            acc[o[0]:o[0] + P[0], o[1]:o[1] + P[1], o[2]:o[2] + P[2]] += s
            cnt[o[0]:o[0] + P[0], o[1]:o[1] + P[1], o[2]:o[2] + P[2]] += 1
    lab_ref = acc.argmax(-1)
    prob_ref = np.moveaxis(acc / cnt[..., None], -1, 0)

    # weights blob of the oracle's variables (format of model.export_weights)
    import struct
    wpath, ipath = str(tmp_path / "net.vnetw"), str(tmp_path / "vol.npy")
    with open(wpath, "wb") as f:
        items = [(k, v.v) for k, v in ps.vars.items()]
        f.write(b"VNETW1\0\0" + struct.pack("<I", len(items)))
        for name, arr in items:
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<I", arr.ndim) + struct.pack("<%dI" % arr.ndim, *arr.shape))
            f.write(arr.tobytes())
    np.save(ipath, vol)
    out = subprocess.run([BIN, "--weights", wpath, "--image", ipath, "--label-out", str(tmp_path / "lab.npy"),
                          "--prob-out", str(tmp_path / "prob.npy"), "--classes", str(K), "--channels", "4",
                          "--levels", str(levels), "--convs", ",".join(map(str, convs)), "--bottom", str(bottom),
                          "--patch", ",".join(map(str, P)), "--stride", ",".join(map(str, stride)), "--batch", str(batch)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-1000:]
    lab, prob = np.load(tmp_path / "lab.npy"), np.load(tmp_path / "prob.npy")
    assert lab.shape == lab_ref.shape and prob.shape == prob_ref.shape
    assert np.abs(prob - prob_ref).max() < 1e-4, np.abs(prob - prob_ref).max()
    margin = np.sort(prob_ref, axis=0)
    sure = (margin[-1] - margin[-2]) > 1e-4
    assert (lab == lab_ref)[sure].all() and (lab == lab_ref).mean() > 0.999


@pytest.mark.parametrize("variant", ["networks", "legacy"])
def test_conv_bias_gradient_closed_form(dev, variant):
    """Every conv bias of the V-Net feeds a train-mode batch-norm, so dLoss/dbias == 0 identically (the batch mean absorbs a
    per-channel shift).  The networks use that closed form (ops.zero_bias_gradients); with it switched off the generic
    column sum of dy must produce nothing but round-off (|db| <= 1e-5 * sum|dy| scale), the oracle's fp64 value is ~1e-12,
    and every OTHER gradient is bit-identical between the two settings."""
    from vnet_tensorflow_amd import networks, VNet, ops, optim
    ps = O.ParamStore(rng=np.random.default_rng(8), perturb=0.2)
    ref_net = O.VNetOracle(3, 0.0, 4, 2, (1, 2), 2, "prelu", variant, ps)
    x, lab = O.synthetic_batch(2, 16, 2, 3, seed=77)
    ref = O.run_step(x.astype(np.float64), lab, ref_net, "sorensen")
    grads = {}
    for fuse in (True, False):
        cls = networks.VNet if variant == "networks" else VNet.VNet
        net = cls(3, 0.0 if variant == "networks" else 1.0, 4, 2, (1, 2), 2, True, "prelu", device=dev)
        net.fuse_zero_bias_grad = fuse
        net.variables.values = {k: v.v for k, v in ps.vars.items()}
        net.build(x.shape)
        flat = optim.FlatParams(net.named_parameters())
        flat.zero_grad()
        logits = net.GetNetwork(g(x, dev)) if variant == "networks" else net.network_fn(g(x, dev))
        loss, _, _, _ = ops.softmax_loss(logits, g(lab, dev, torch.int32), "sorensen")
        loss.backward()
        torch.cuda.synchronize()
        grads[fuse] = {n: p.grad.detach().cpu().numpy().copy() for n, p in net.named_parameters()}
    nb = 0
    for n in grads[True]:
        if n.endswith("biases"):
            nb += 1
            assert np.abs(ref["grads"][n]).max() < 1e-9, (n, np.abs(ref["grads"][n]).max())       # oracle: zero up to fp64 round-off
            if not n.startswith("vnet/output_layer"):                                              # (the 1x1x1 head kernel computes db with dw)
                assert np.abs(grads[True][n]).max() == 0.0, n                                      # closed form: exact 0
            assert np.abs(grads[False][n]).max() < 1e-4, (n, np.abs(grads[False][n]).max())         # generic path: round-off only
        else:
            assert np.array_equal(grads[True][n], grads[False][n]), n
    assert nb >= 8


@pytest.mark.parametrize("compute", ["fp32"])
def test_gradient_accumulation_in_the_producing_kernel(dev, compute):
    """Tensors with two consumers (skip connection, residual block input): the second gradient is added by the backward-data
    kernel that produces it (y += ..., ops.fork) instead of an autodiff add kernel.  Same two fp32 numbers are added either
    way, so every gradient is bit-identical to the unfused graph; and the fused pass really runs no add kernel."""
    from vnet_tensorflow_amd import networks, ops, optim
    x, lab = O.synthetic_batch(1, 32, 2, 3, seed=5)
    grads = {}
    ops.set_compute_dtype(compute)
    try:
        for fuse in (True, False):
            np.random.seed(11)
            net = networks.VNet(3, 0.0, 8, 3, (1, 2, 3), 2, True, "prelu", device=dev)
            net.fuse_grad_accumulation = fuse
            net.build(x.shape)
            flat = optim.FlatParams(net.named_parameters())
            flat.zero_grad()
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU]) as prof:
                loss, _, _, _ = ops.softmax_loss(net.GetNetwork(g(x, dev)), g(lab, dev, torch.int32), "sorensen")
                loss.backward()
            torch.cuda.synchronize()
            adds = sum(1 for e in prof.events() if e.name == "aten::add" and e.input_shapes != [])
            big_adds = [e for e in prof.events() if e.name in ("aten::add", "aten::add_")]
            grads[fuse] = ({n: p.grad.detach().cpu().numpy().copy() for n, p in net.named_parameters()}, len(big_adds))
    finally:
        ops.set_compute_dtype("fp32")
    for n in grads[True][0]:
        assert np.array_equal(grads[True][0][n], grads[False][0][n]), n
    # 3 levels: 3 skip forks + residual forks at levels 2, 3 and the bottom = 6 full-tensor adds saved
    assert grads[False][1] - grads[True][1] >= 6, (grads[False][1], grads[True][1])


@pytest.mark.parametrize("mode,ks,stride,shape,Cin,Cout,residual", [
    ("fp32", 5, 1, (1, 32, 32, 32), 16, 16, True),        # one brick row per workgroup, residual added in the epilogue
    ("fp32", 5, 1, (2, 24, 20, 28), 8, 24, False),        # ragged bricks: voxels outside the volume must not be counted
    ("fp32", 5, 1, (1, 8, 8, 8), 64, 64, True),           # split-K: statistics come from the reduce kernel
    ("fp32", 2, 2, (1, 32, 32, 32), 16, 32, False),       # 2^3 stride-2 down convolution
])
def test_batch_norm_statistics_from_the_conv_epilogue(dev, mode, ks, stride, shape, Cin, Cout, residual):
    """The convolution writes per-workgroup partial sums of y (+ residual) and its square; the batch-norm behind it only
    finalizes them (vnet_conv_fwd_stats + vnet_bn_finalize_partial).  Mean / inverse standard deviation must equal the
    float64 moments of the tensor the batch-norm normalises, the normalised output must equal the unfused path's, and the
    moving averages must be updated."""
    from vnet_tensorflow_amd import ops
    gen = torch.Generator().manual_seed(Cin * 7 + Cout)
    B, D, H, W = shape
    x = (torch.randn(B, D, H, W, Cin, generator=gen) * 1.5 + 0.3).to(dev)
    w = (torch.randn(ks, ks, ks, Cin, Cout, generator=gen) * 0.05).to(dev)
    b = torch.randn(Cout, generator=gen).to(dev)
    od = tuple(-(-v // stride) for v in (D, H, W))
    r = (torch.randn(B, *od, Cout, generator=gen) * 2.0).to(dev) if residual else None
    gamma, beta = (torch.rand(Cout, generator=gen) + 0.5).to(dev), torch.randn(Cout, generator=gen).to(dev)
    ops.set_compute_dtype(mode)
    try:
        outs = {}
        for fused in (True, False):
            ops.set_epilogue_bn_stats(fused, fp32_direct=True)
            mm, mv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
            with torch.no_grad():
                y = ops.conv(x, w, b, ks, stride, bn_stats=True, bn_residual=r)
                assert (getattr(y, "_vnet_stats", None) is not None) == fused
                z, mean, invstd = ops.bn_act(y, gamma, beta, "relu", None, r, False, mm, mv, want_stats=True)
            outs[fused] = (y.clone(), z.clone(), mean.clone(), invstd.clone(), mm.clone(), mv.clone())
    finally:
        ops.set_epilogue_bn_stats(True, fp32_direct=True)
        ops.set_compute_dtype("fp32")
    yf, zf, mean, invstd, mm, mv = outs[True]
    assert torch.equal(yf, outs[False][0])                                   # the convolution output itself is unchanged
    s = yf.double() + (r.double() if residual else 0.0)
    mu = s.mean(dim=(0, 1, 2, 3))
    var = s.var(dim=(0, 1, 2, 3), unbiased=False)
    check_close("mean", mean, mu.cpu().numpy(), 1e-6, atol=1e-6)
    check_close("invstd", invstd, (1.0 / torch.sqrt(var + 1e-3)).cpu().numpy(), 1e-6)
    check_close("normalised output", zf, outs[False][1].cpu().numpy(), 2e-6)
    check_close("moving mean", mm, (0.01 * mu).cpu().numpy(), 1e-5, atol=1e-7)
    check_close("moving variance", mv, (0.99 + 0.01 * var).cpu().numpy(), 1e-6)


@pytest.mark.parametrize("taps,I,O", [(125, 16, 16), (125, 32, 16), (125, 4, 16), (125, 24, 40), (8, 16, 32), (125, 6, 10),
                                      (125, 32, 32), (125, 64, 32), (125, 32, 96)])      # whole 32-channel blocks: both bf16 / fp32 / f32x3 images from one read
def test_batched_filter_repack_equals_single_pack(dev, taps, I, O):
    """The one-launch repack of every registered filter (after each optimiser step) must produce, bit for bit, the images
    the single-filter packer produces -- all layouts (fp32 forward / backward-data / transposed, bf16 forward / backward, f32x3 forward /
    backward: the three exactly split bf16 pieces)."""
    from vnet_tensorflow_amd import ops
    ks = 5 if taps == 125 else 2
    gen = torch.Generator().manual_seed(taps + I + O)
    w = torch.nn.Parameter(torch.randn(ks, ks, ks, I, O, generator=gen).to(dev))
    modes = [ops.PACK_FWD, ops.PACK_BWD] + ([ops.PACK_FWD_BF16, ops.PACK_BWD_BF16, ops.PACK_FWD_X3, ops.PACK_BWD_X3] if taps == 125 else [])
    ops.clear_pack_registry()
    try:
        single = {m: ops.packed_weights(w, m, taps, I, O).clone() for m in modes}          # registers (w, mode) and packs one by one
        for m in modes:
            w._vnet_packed[(m, taps, I, O)][1].fill_(float("nan"))                             # wipe, then refresh all in one launch
        with torch.no_grad():
            w.mul_(1.0)
        ops.invalidate_packed()
        ops.repack_registered()
        for m in modes:
            got = w._vnet_packed[(m, taps, I, O)][1]
            assert torch.equal(got.view(torch.int32), single[m].view(torch.int32)), (m, taps, I, O)
    finally:
        ops.clear_pack_registry()


@pytest.mark.parametrize("compute,cin", [("fp32", 1), ("fp32", 2), ("bf16", 4)])
def test_deferred_batched_filter_gradient_reduce(dev, compute, cin):
    """ops.deferred_wgrad_reduce: the filter-gradient launches leave their partial slabs in per-layer buffers and ONE launch
    reduces all of them at the end of the backward pass (vnet_wgrad_defer / vnet_wgrad_flush).  Same summation order as the
    per-layer reduce: every gradient bit-identical; the queue really fills and really drains; the fused input block's G
    (consumed inside backward) is still reduced on the spot."""
    from vnet_tensorflow_amd import networks, ops, optim
    from vnet_tensorflow_amd._lib import lib
    L = lib()
    x, lab = O.synthetic_batch(1, 32, cin, 3, seed=5)
    grads, queued = {}, {}
    ops.set_compute_dtype(compute)
    # (round 4: a deferring pass in bf16 storage ALSO launches its deep-level filter gradients as one group, which splits the layers
    #  differently -- another summation order; that path has its own tests in tests/test_hip_wgrad_group.py.  Here: the reduce.)
    ops.set_wgrad_group(False)
    try:
        for defer in (True, False):
            np.random.seed(11)
            net = networks.VNet(3, 0.0, 16, 3, (1, 2, 3), 2, True, "prelu", device=dev)
            net.build(x.shape)
            flat = optim.FlatParams(net.named_parameters())
            for rep in range(2):                   # second pass: the per-layer slab buffers are reused
                flat.zero_grad()
                loss, _, _, _ = ops.softmax_loss(net.GetNetwork(g(x, dev)), g(lab, dev, torch.int32), "sorensen")
                with ops.deferred_wgrad_reduce(defer):
                    loss.backward()
                    ops.join_param_grad_stream()
                    queued[defer] = L.vnet_wgrad_pending(ops._stream())
                assert L.vnet_wgrad_pending(ops._stream()) == 0
            torch.cuda.synchronize()
            grads[defer] = {n: p.grad.detach().cpu().numpy().copy() for n, p in net.named_parameters()}
    finally:
        ops.set_compute_dtype("fp32")
        ops.set_wgrad_group(True)
    assert queued[False] == 0 and queued[True] >= 10, queued
    for n in grads[True]:
        assert np.array_equal(grads[True][n], grads[False][n]), n
    assert all(np.isfinite(v).all() for v in grads[True].values())

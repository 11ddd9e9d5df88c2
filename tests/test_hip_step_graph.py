"""-m gpu: the whole training step as ONE hipGraph (model.image2label.train_step; reference model.py:743-748 is one
sess.run per step).  The graph holds exactly the launches the eager step makes -- forward, loss, backward on two
streams, fused optimiser, batched filter repack -- with the per-step scalars (learning rate, Adam's lr_t, dropout
stream position) read from the device step state, so replay must be BIT-identical to the eager step."""
import os
import pathlib
import socket
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _cfg(P=16, dropout=0.0, opt="Adam", compute="fp32", cin=1, K=2, loss="sorensen"):
    return {"TrainingSetting": {
        "Data": {"TrainingDataDirectory": "synthetic", "TestingDataDirectory": "synthetic",
                 "ImageFilenames": ["image%d.npy" % i for i in range(cin)], "LabelFilename": "label.npy",
                 "Synthetic": {"Cases": 4}},
        "SegmentationClasses": list(range(K)), "BatchSize": 2, "PatchShape": [P] * 3, "ComputeDtype": compute,
        "Networks": {"Name": "VNet", "Dropout": dropout, "NumChannel": 8, "NumLevels": 3, "NumConvolutions": [1, 2, 2],
                     "BottomConvolutions": 2},
        "Loss": {"Name": loss, "Weights": [0.3, 0.7, 1.0, 0.5, 0.2][:K], "Alpha": 0.5},
        "Optimizer": {"Name": opt, "InitialLearningRate": 1e-2, "Momentum": 0.9, "Decay": {"Factor": 0.9, "Steps": 3}}}}


def _run(dev, graph, steps, monkeypatch, **kw):
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    monkeypatch.setenv("VNET_STEP_GRAPH", "1" if graph else "0")
    cfg = _cfg(**kw)
    T = cfg["TrainingSetting"]
    cin, K, P = len(T["Data"]["ImageFilenames"]), len(T["SegmentationClasses"]), T["PatchShape"][0]
    np.random.seed(7)
    m = image2label(None, cfg, device=dev, verbose=False)
    try:
        m.read_config()
        m.build_model_graph()
        m._setup_training()
        batches = [synthetic_batch(2, P, cin, K, seed=40 + i) for i in range(2)]
        batches = [(torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)) for x, l in batches]
        losses = []
        for i in range(steps):
            x, l = batches[i % 2]                      # the inputs change between replays: static-buffer copy is exercised
            losses.append(float(m.train_step(x, l)))
        torch.cuda.synchronize()
        assert (m._graph_mode() == "whole") == graph
        if graph:
            assert m._graphs is not None and len(m._graphs) == 1, "the step was never captured"
    finally:
        ops.set_compute_dtype("fp32")
    state = {k: v.clone() for k, v in m.network.state_dict().items()}
    return losses, m.flat.data.clone(), state, {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in m.optimizer.state_dict().items()}


@pytest.mark.parametrize("kw", [
    dict(),                                                       # Adam, no dropout (the bench configuration)
    dict(dropout=0.05),                                           # dropout masks from the device step state
    dict(opt="NesterovMomentum", loss="mixed_weighted_jaccard", cin=2, K=3),
    dict(opt="SGD", compute="bf16", cin=4, K=5),                  # BASELINE config C5 arithmetic
], ids=["adam", "dropout", "nesterov-mixed", "sgd-bf16"])
def test_graph_replay_is_bit_identical_to_eager(dev, monkeypatch, kw):
    """6 steps: 2 eager warm-up steps + capture + 4 replays against 6 eager steps -- same losses, parameters, moving
    statistics and optimiser slots, bit for bit (LR decays every step, so a frozen scalar would show)."""
    a = _run(dev, True, 6, monkeypatch, **kw)
    b = _run(dev, False, 6, monkeypatch, **kw)
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    for k in a[3]:
        assert (torch.equal(a[3][k], b[3][k]) if isinstance(a[3][k], torch.Tensor) else a[3][k] == b[3][k]), k
    assert a[0][-1] < a[0][0]                                      # and it trains


def test_graph_step_is_one_host_call(dev, monkeypatch):
    """Host cost of a replayed step: a handful of microsecond-scale calls instead of ~400 ctypes launches."""
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    res = {}
    for graph in (True, False):
        monkeypatch.setenv("VNET_STEP_GRAPH", "1" if graph else "0")
        np.random.seed(7)
        m = image2label(None, _cfg(P=32), device=dev, verbose=False)
        m.read_config()
        m.build_model_graph()
        m._setup_training()
        x, l = synthetic_batch(2, 32, 1, 2, seed=40)
        x, l = torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)
        for _ in range(4):
            m.train_step(x, l)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.train_step(x, l)
        res[graph] = (time.perf_counter() - t0) / 10
        torch.cuda.synchronize()
    print("host enqueue per step: graph %.3f ms, eager %.3f ms" % (res[True] * 1e3, res[False] * 1e3))
    assert res[True] < 0.5 * res[False] or res[True] < 1e-3


def test_timed_launches_next_to_the_graph(dev, monkeypatch):
    """bench.py's roofline leg: launches cannot be timed inside a replayed hipGraph on this runtime, so a capture made
    while profiling is on must simply carry no events, and the same model switched to eager enqueue (force_eager) times
    the tagged launches with HIP events."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    monkeypatch.setenv("VNET_STEP_GRAPH", "1")
    np.random.seed(7)
    m = image2label(None, _cfg(P=32), device=dev, verbose=False)
    m.read_config()
    m.build_model_graph()
    m._setup_training()
    x, l = synthetic_batch(2, 32, 1, 2, seed=40)
    x, l = torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)
    fam = {"conv k5 s1 32^3x2 16->8", "wgrad k5 s1 32^3x2 16->8"}
    ops.profile_start(fam)
    try:
        for _ in range(4):                    # 2 eager (timed) + capture (untimed) + replay
            m.train_step(x, l)
        assert m._graphs is not None
        assert len(ops.profile_stop()) == 2 * len(fam)
        m.force_eager = True
        ops.profile_start(fam)
        for _ in range(3):
            m.train_step(x, l)
        recs = ops.profile_stop()
    finally:
        ops.profile_stop()
    assert sorted(set(t for t, _, _, _ in recs)) == sorted(fam) and len(recs) == 3 * len(fam), recs
    for tag, fl, by, ms in recs:
        assert 0.0 < ms < 5.0, (tag, ms)
        assert fl == 2.0 * 2 * 32 ** 3 * 125 * 16 * 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out, mode):
    """One rank, process group "nccl" (= RCCL) on cuda:0, the data-parallel collective path forced on."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      VNET_DP_FORCE="1", VNET_DP_BUCKET_BYTES=str(16 << 10))
    os.environ["VNET_STEP_GRAPH"] = "0" if mode.startswith("eager") else "1"
    os.environ["VNET_DP_TWO_PASS"] = "0" if mode.endswith("1p") else "1"
    if mode == "serial":
        os.environ["VNET_DP_MODE"] = "serial"
    import torch.distributed as dist
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    np.random.seed(7)
    import json
    kw = json.loads(os.environ.get("VNET_TEST_DP_CFG", "{}"))
    m = image2label(None, _cfg(**kw), device=dev, verbose=False)
    m.read_config()
    m.build_model_graph()
    m._setup_training()
    assert m.sync is not None and m.sync.active and len(m.sync.buckets) >= 3
    x, l = synthetic_batch(2, 16, kw.get("cin", 1), kw.get("K", 2), seed=40)
    x, l = torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)
    losses = [float(m.train_step(x, l)) for _ in range(5)]
    torch.cuda.synchronize()
    assert m._graph_mode() == ("off" if mode.startswith("eager") else "segmented")
    if mode == "serial":
        assert m.step_mode() == "serial" and m.sync.launch_log and len(m.sync.launch_log) == len(m.sync.buckets)
    assert m._two_pass == (not mode.endswith("1p"))
    if not mode.startswith("eager"):
        # gradients graph 1 (forward + decoder / bottom backward), [gradients graph 2 (encoder backward)], optimiser graph
        assert m._graphs is not None and len(m._graphs) == (2 if mode.endswith("1p") else 3)
        if not mode.endswith("1p"):
            # the first pass's buckets end exactly where the bottom level's gradients end
            names = m.flat.names
            k = m.sync.phase1_last
            assert all(not n.startswith(("vnet/encoder", "vnet/input_layer")) for n in names[:k + 1])
            assert all(n.startswith(("vnet/encoder", "vnet/input_layer")) for n in names[k + 1:])
            assert any(last == k + 1 for _, _, _, last in m.sync.buckets)
    torch.save({"losses": losses, "data": m.flat.data.cpu()}, os.path.join(out, "dp_%s.pt" % mode))
    dist.destroy_process_group()


def test_data_parallel_segmented_graph_rccl_group_of_one(tmp_path, dev):
    """Data-parallel step with RCCL in the loop (group of one rank: all a 1-GPU box can host).
    eager      : kernel-by-kernel enqueue, bucket all-reduces launched from the gradient hooks (overlapping backward);
    segmented  : gradients graph 1 (forward, decoder + bottom-level backward) -> all-reduce of those buckets, asynchronous
                 -> gradients graph 2 (encoder backward) -> remaining buckets -> optimiser graph;
    serial     : the segmented graphs replayed with every all-reduce after the second gradients graph (no collective shares
                 the CUs with backward);
    *1p        : the same with a single backward pass (VNET_DP_TWO_PASS=0: gradients graph -> all buckets -> optimiser).
    No collective is captured (a captured RCCL all-reduce trips ProcessGroupNCCL's watchdog: hipErrorCapturedEvent).
    All five must produce the same losses and parameters bit for bit (the cut adds the two gradients of a skip tensor with
    an add kernel instead of in the backward-data epilogue: the same two fp32 numbers)."""
    modes = ("eager", "eager1p", "segmented", "serial", "segmented1p")
    for md in modes:
        mp.spawn(_dp_worker, args=(1, _free_port(), str(tmp_path), md), nprocs=1, join=True)
    a = torch.load(tmp_path / "dp_eager.pt")
    for md in modes[1:]:
        b = torch.load(tmp_path / ("dp_%s.pt" % md))
        assert a["losses"] == b["losses"], md
        assert torch.equal(a["data"], b["data"]), md


def test_data_parallel_segmented_graph_bf16_storage(tmp_path, dev, monkeypatch):
    """The same with bf16 tensors end to end (BASELINE config C5's per-GPU arithmetic): graph replay == eager enqueue and
    segmented == serial bit for bit; the two-pass backward differs from the one-pass form only in where the second gradient of a
    cut tensor is rounded (sum of two bf16 tensors instead of one rounding of bf16 + fp32 accumulator), i.e. by bf16 round-off."""
    monkeypatch.setenv("VNET_TEST_DP_CFG", '{"opt": "SGD", "compute": "bf16", "cin": 4, "K": 5}')
    # Round 4: step forms that defer their filter-gradient reduces (graph replays, the serial step) also launch the deep-level filter
    # gradients as ONE group, which splits a layer over fewer workgroups -- another summation order than the eager data-parallel step,
    # whose gradients must leave from the hooks while backward runs.  Bit identity across ALL step forms is therefore a property of
    # VNET_WGRAD_GROUP=0 (first leg); with the grouped launch (the default) the deferring forms agree with each other bit for bit
    # and with the eager step to summation order.
    monkeypatch.setenv("VNET_WGRAD_GROUP", "0")
    modes = ("eager1p", "segmented1p", "eager", "segmented", "serial")
    for md in modes:
        mp.spawn(_dp_worker, args=(1, _free_port(), str(tmp_path), md), nprocs=1, join=True)
    r = {md: torch.load(tmp_path / ("dp_%s.pt" % md)) for md in modes}
    for a, b in (("eager1p", "segmented1p"), ("eager", "segmented"), ("segmented", "serial")):
        assert r[a]["losses"] == r[b]["losses"], (a, b)
        assert torch.equal(r[a]["data"], r[b]["data"]), (a, b)
    monkeypatch.delenv("VNET_WGRAD_GROUP")
    grouped = {}
    for md in ("segmented", "serial"):
        mp.spawn(_dp_worker, args=(1, _free_port(), str(tmp_path), md), nprocs=1, join=True)
        grouped[md] = torch.load(tmp_path / ("dp_%s.pt" % md))
    assert grouped["segmented"]["losses"] == grouped["serial"]["losses"] and torch.equal(grouped["segmented"]["data"], grouped["serial"]["data"])
    dg = (grouped["segmented"]["data"] - r["segmented"]["data"]).norm() / r["segmented"]["data"].norm()
    assert 0.0 < float(dg) < 1e-3, float(dg)               # (not zero: the grouped launch really ran)
    l1, l2 = np.array(r["eager1p"]["losses"]), np.array(r["eager"]["losses"])
    assert np.all(np.isfinite(l1)) and np.all(np.isfinite(l2)) and np.abs(l1 - l2).max() < 2e-2 * np.abs(l1).max(), (l1, l2)
    d = (r["eager1p"]["data"] - r["eager"]["data"]).norm() / r["eager"]["data"].norm()
    assert float(d) < 2e-2, float(d)


def test_refused_capture_falls_back_to_eager_steps(dev, monkeypatch):
    """A capture the runtime refuses (RuntimeError from torch.cuda.graph) must not end the job: the step keeps running
    eagerly -- same math (the bit-identity tests above) -- and never retries the capture."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    monkeypatch.setenv("VNET_STEP_GRAPH", "1")
    np.random.seed(7)
    m = image2label(None, _cfg(), device=dev, verbose=False)
    m.read_config()
    m.build_model_graph()
    m._setup_training()
    calls = []

    def refuse(*a, **k):
        calls.append(1)
        raise RuntimeError("hipErrorStreamCaptureUnsupported (simulated)")
    monkeypatch.setattr(m, "_build_step_graph", refuse)
    x, l = synthetic_batch(2, 16, 1, 2, seed=40)
    x, l = torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)
    losses = [float(m.train_step(x, l)) for _ in range(6)]
    assert len(calls) == 1 and m.step_mode() == "off" and m.global_step == 6
    ref, _, _, _ = _run(dev, False, 6, monkeypatch)
    # same seed, same first batch repeated vs alternating batches: only the first step is comparable
    assert losses[0] == ref[0] and all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("kw", [dict(), dict(opt="SGD", compute="bf16", cin=4, K=5)], ids=["fp32", "c5-bf16"])
def test_step_without_gradient_memset_never_reads_stale_gradients(dev, monkeypatch, kw):
    """FlatParams.begin_step() replaces the per-step memset of the gradient buffer, which is sound only if every
    gradient of a step is WRITTEN before anything adds to it.  Here every slice a backward kernel wrote is poisoned with NaN
    between steps: a gradient that was accumulated onto last step's value (instead of written) would turn into NaN.  Losses
    and parameters must equal, bit for bit, the run that clears the buffer every step; the slices nobody writes (conv biases in
    front of batch-norms, dead batch-norms, padding) must stay exactly zero; and the guard itself must notice an autograd
    accumulation (p.grad += g)."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    monkeypatch.setenv("VNET_STEP_GRAPH", "0")
    cfg = _cfg(**kw)
    T = cfg["TrainingSetting"]
    cin, K, P = len(T["Data"]["ImageFilenames"]), len(T["SegmentationClasses"]), T["PatchShape"][0]
    x, l = synthetic_batch(2, P, cin, K, seed=40)
    out = {}
    try:
        for mode in ("poisoned", "memset"):
            np.random.seed(7)
            m = image2label(None, cfg, device=dev, verbose=False)
            m.read_config(); m.build_model_graph(); m._setup_training()
            xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)
            m.flat.needs_zero = mode == "memset"
            losses, untouched = [], None
            for step in range(5):
                losses.append(float(m.train_step(xt, lt)))
                written = [p._vnet_sink.written for p in m.flat.params]
                if mode == "poisoned":
                    assert not m.flat.needs_zero, "the networks' own backward must not accumulate through autograd"
                    assert torch.isfinite(m.flat.grad).all()
                    for n, p, w in zip(m.flat.names, m.flat.params, written):
                        if w and not n.endswith("biases"):      # (conv biases: closed form, "written" without a kernel write)
                            p.grad.fill_(float("nan"))
                    untouched = [p for p, w in zip(m.flat.params, written) if not w]
            torch.cuda.synchronize()
            out[mode] = (losses, m.flat.data.clone())
            if mode == "poisoned":
                assert sum(written) > 20 and all(float(p.grad.abs().max()) == 0.0 for p in untouched)
    finally:
        ops.set_compute_dtype("fp32")
    assert out["poisoned"][0] == out["memset"][0], (out["poisoned"][0], out["memset"][0])
    assert torch.equal(out["poisoned"][1], out["memset"][1])
    # the guard: an in-place autograd accumulation into the buffer is noticed (first step of a buffer: cleared anyway)
    from vnet_tensorflow_amd import optim
    p = torch.nn.Parameter(torch.ones(8, device=dev))
    flat = optim.FlatParams([("p", p)])
    flat.begin_step()
    (p * 2.0).sum().backward()                     # plain autograd: AccumulateGrad adds into the flat buffer's view
    flat.check_accumulation()
    assert flat.needs_zero and torch.equal(p.grad, torch.full((8,), 2.0, device=dev))
    flat.begin_step()
    (p * 3.0).sum().backward()
    assert torch.equal(p.grad, torch.full((8,), 3.0, device=dev))      # cleared first: not 5

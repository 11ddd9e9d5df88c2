"""CPU, world_size 2 over gloo: the data-parallel exchange step (bucketed gradient all-reduce launched
from autograd hooks into the flat gradient buffer, rank-0 parameter broadcast, 1/world scaling) on a
small stand-in model -- exactly the code path the RCCL run uses, minus the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [("l1/weights", (6, 5)), ("l1/biases", (5,)), ("dead/gamma", (5,)), ("l2/weights", (5, 3)), ("l2/biases", (3,))]
    return [(n, torch.nn.Parameter(torch.randn(s, generator=g))) for n, s in shapes]


def _loss(params, x):
    p = dict(params)
    h = torch.tanh(x @ p["l1/weights"] + p["l1/biases"])
    return ((h @ p["l2/weights"] + p["l2/biases"]) ** 2).mean()      # "dead/gamma" never gets a gradient


def _worker(rank, world, port, out, hold=0.0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vnet_tensorflow_amd import optim, parallel
    r, _, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    params = _model(seed=100 + rank)                       # different init per rank ...
    flat = optim.FlatParams(params)
    parallel.broadcast_parameters(flat.data)               # ... until rank 0's weights are broadcast
    sync = parallel.BucketedGradAllReduce(flat, bucket_bytes=64, hold_fraction=hold)
    assert len(sync.buckets) >= 2
    results = []
    for step in range(2):
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + rank))
        flat.zero_grad()
        sync.begin_step()
        _loss(params, x).backward()
        if step == 1:            # (step 0 calibrates the event counts: nothing is launched before finish())
            if hold > 0.0:       # held buckets go out together, in order, once `hold` of the bytes is ready -- or in finish()
                n = sum(sync._launched)
                assert n == 0 or sync._ready_bytes >= hold * sync._total_bytes
                assert sync._launched == sorted(sync._launched, reverse=True) or n == 0
            else:
                assert sum(sync._launched) >= 1
        sync.finish()
        assert all(sync._launched) and not sync._held
        results.append(flat.grad.clone())
    torch.save({"data": flat.data.clone(), "grads": results}, os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_bucketed_allreduce_world4(tmp_path):
    """Four ranks (the driver scales to 8): every rank ends with the sum of the four per-rank gradients."""
    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), 0.6), nprocs=4, join=True)
    r = [torch.load(tmp_path / ("r%d.pt" % k)) for k in range(4)]
    from vnet_tensorflow_amd import optim
    params = _model(seed=100)
    flat = optim.FlatParams(params)
    for step in range(2):
        tot = torch.zeros_like(flat.grad)
        for rank in range(4):
            x = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + rank))
            flat.zero_grad()
            _loss(params, x).backward()
            tot += flat.grad
        for k in range(4):
            assert torch.allclose(r[k]["grads"][step], tot, atol=1e-6)
            assert torch.equal(r[k]["data"], r[0]["data"])


@pytest.mark.parametrize("hold", [0.0, 0.6, 1.0])
def test_bucketed_allreduce_world2(tmp_path, hold):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), hold), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["data"], r1["data"])             # broadcast made the replicas identical
    # single-process reference: sum of the two per-rank gradients
    from vnet_tensorflow_amd import optim
    params = _model(seed=100)
    flat = optim.FlatParams(params)
    for step in range(2):
        tot = torch.zeros_like(flat.grad)
        for rank in range(2):
            x = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + rank))
            flat.zero_grad()
            _loss(params, x).backward()
            tot += flat.grad
        assert torch.allclose(r0["grads"][step], tot, atol=1e-6)
        assert torch.equal(r0["grads"][step], r1["grads"][step])
    # the never-touched ("dead") variable stays exactly zero after the flushed all-reduce
    o = flat.offsets[flat.names.index("dead/gamma")]
    assert float(r0["grads"][0][o:o + 5].abs().sum()) == 0.0


# ---- launch point of the held buckets (the threshold must not depend on where the size cuts fall) ----------
def _chain_model(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [("a", (6, 6)), ("b", (6, 6)), ("c", (6, 40)), ("d", (40, 10)), ("e", (10, 2))]     # forward order
    return [(n, torch.nn.Parameter(0.3 * torch.randn(s, generator=g))) for n, s in shapes]


def _chain_loss(params, x):
    h = x
    for _, p in params:
        h = torch.tanh(h @ p)
    return (h ** 2).mean()


def _hold_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vnet_tensorflow_amd import optim, parallel
    parallel.init_from_env("gloo")
    params = _chain_model(5)
    flat = optim.FlatParams(params)          # gradient-production order: e (20), d (400), c (240), b (36), a (36) floats
    # one size-driven bucket would hold everything; 55 % of the bytes exist once d's gradient is written
    sync = parallel.BucketedGradAllReduce(flat, bucket_bytes=1 << 20, hold_fraction=0.55)
    assert [(f, l) for _, _, f, l in sync.buckets] == [(0, 2), (2, 5)], sync.buckets
    logs = []
    for step in range(3):
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(step))
        flat.zero_grad()
        sync.begin_step()
        _chain_loss(params, x).backward()
        logs.append(list(sync.launch_log))
        sync.finish()
    torch.save({"logs": logs, "grad": flat.grad.clone()}, os.path.join(out, "h%d.pt" % rank))
    dist.destroy_process_group()


def test_hold_fraction_launches_before_backward_ends(tmp_path):
    """With hold_fraction = f the collective starts when a fraction f of the gradient BYTES exists -- here after 2 of the 5
    gradient events -- not when the last size-driven bucket completes (which would be the end of backward)."""
    mp.spawn(_hold_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = torch.load(tmp_path / "h0.pt")
    assert r["logs"][0] == []                                 # calibration step: nothing before finish()
    for log in r["logs"][1:]:
        assert log[0] == (0, 2), log                           # bucket 0 went out after the 2nd of 5 gradient events
        assert log[-1] == (1, 5)
    assert torch.equal(r["grad"], torch.load(tmp_path / "h1.pt")["grad"])


# ---- dataset sharding (unequal step counts hang the collectives) ------------------------------------
def _shard_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vnet_tensorflow_amd import data, parallel
    parallel.init_from_env("gloo")
    ds = data.VolumeDataset("synthetic", ["image.npy"], "label.npy", [0, 1], (8, 8, 8), 1, train=True, seed=3,
                            synthetic={"Cases": 10, "Shape": [10, 10, 10]}, rank=rank, world=world)
    seen = []
    for epoch in range(2):
        n = 0
        plan = ds.epoch_plan()
        for cases, seeds in plan:
            img, lab = ds.make_batch(cases, seeds)
            assert img.shape == (1, 8, 8, 8, 1) and lab.dtype == np.int32
            t = torch.ones(1)
            dist.all_reduce(t)            # the per-step collective: pairs up only if every rank takes the same number of steps
            assert float(t) == world
            n += 1
            seen.append((epoch, cases[0]))
        assert n == ds.steps_per_epoch() == 10 // world
    torch.save(seen, os.path.join(out, "d%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 4])
def test_dataset_shards_are_disjoint_and_equal_length(tmp_path, world):
    """10 cases on 3 / 4 ranks (10 % world != 0): every rank runs floor(10/world) steps per epoch, no case is seen twice in
    an epoch, and the per-step collective never waits for a rank that has already finished."""
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seen = [torch.load(tmp_path / ("d%d.pt" % r)) for r in range(world)]
    for epoch in range(2):
        cases = [c for s in seen for (e, c) in s if e == epoch]
        assert len(cases) == world * (10 // world) and len(set(cases)) == len(cases)
    assert [c for s in seen for (e, c) in s if e == 0] != [c for s in seen for (e, c) in s if e == 1]     # reshuffled per epoch


def test_dataset_refuses_a_rank_without_work():
    from vnet_tensorflow_amd import data
    with pytest.raises(ValueError):
        data.VolumeDataset("synthetic", ["image.npy"], "label.npy", [0, 1], (8, 8, 8), 2, synthetic={"Cases": 3}, rank=0, world=2)


def test_test_dataset_is_not_sharded():
    """the TEST pass holds no collective, so a test set smaller than world x batch must not abort data-parallel
    training, and no test case may be dropped because of the rank count (reference model.py:289-295: one unsharded pipeline)."""
    from vnet_tensorflow_amd import data
    plans = []
    for rank in range(8):
        ds = data.VolumeDataset("synthetic", ["image.npy"], "label.npy", [0, 1], (8, 8, 8), 1, train=False,
                                synthetic={"Cases": 4}, rank=rank, world=8)            # 4 cases, 8 ranks: used to raise
        assert ds.steps_per_epoch() == 4
        plans.append([c for cases, _ in ds.epoch_plan() for c in cases])
    assert all(p == [0, 1, 2, 3] for p in plans), plans
    ds = data.VolumeDataset("synthetic", ["image.npy"], "label.npy", [0, 1], (8, 8, 8), 2, train=False, synthetic={"Cases": 5}, rank=1, world=2)
    assert [c for cases, _ in ds.epoch_plan() for c in cases] == [0, 1, 2, 3]          # drop_remainder, like the reference


def test_prefetcher_keeps_order_and_content():
    from vnet_tensorflow_amd import data
    mk = lambda: data.VolumeDataset("synthetic", ["image.npy"], "label.npy", [0, 1, 2], (8, 8, 8), 2, train=True, seed=1,
                                    synthetic={"Cases": 9, "Shape": [12, 10, 9]})
    plain = list(mk())
    pre = list(data.Prefetcher(mk(), depth=3, workers=3, pin=False))
    assert len(plain) == len(pre) == 4
    for (a, b), (c, d) in zip(plain, pre):
        assert np.array_equal(a, c.numpy()) and np.array_equal(b, d.numpy())


# ---- start-up choice of the data-parallel step mode ---------------------------------------------------------------------
def _tune_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import time
    from vnet_tensorflow_amd import parallel
    parallel.init_from_env("gloo")
    cost = {"segmented": (0.003, 0.003), "off": (0.001, 0.012)}       # seconds per step on (rank 0, rank 1)
    tuner = parallel.StepModeAutotune(["segmented", "off"], steps=4, blocks=3)
    assert tuner.total_steps() == 24
    seen = []
    for step in range(28):
        mode = tuner.mode()
        tuner.before()
        # one NOISY block: the second 'off' block is fast on both ranks (a single 4-step block must not decide the job)
        fast = mode == "off" and 12 <= step < 16
        time.sleep(0.0005 if fast else cost[mode][rank])
        seen.append(mode)
        tuner.after()
    torch.save({"seen": seen, "choice": tuner.choice, "times": tuner.times, "samples": tuner.samples}, os.path.join(out, "t%d.pt" % rank))
    dist.destroy_process_group()


def test_step_mode_autotune_picks_the_same_winner_on_every_rank(tmp_path):
    """Rank 0 alone would prefer 'off' (1 ms vs 3 ms per step), but rank 1 is slow in that mode (12 ms): the max over ranks
    decides.  The candidates are measured in three interleaved rounds and scored by their MEDIAN block, so one lucky block of
    'off' (0.5 ms per step on both ranks) does not flip the choice; both ranks continue with 'segmented'."""
    mp.spawn(_tune_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "t0.pt"), torch.load(tmp_path / "t1.pt")
    assert r0["seen"] == r1["seen"] == (["segmented"] * 4 + ["off"] * 4) * 3 + ["segmented"] * 4
    assert r0["choice"] == r1["choice"] == "segmented" and r0["times"] == r1["times"] and r0["samples"] == r1["samples"]
    assert min(r0["samples"][1]) < r0["times"][0] < r0["times"][1]          # the lucky block was the fastest of all, the median is not


def test_step_mode_autotune_prefers_serial_within_the_margin():
    """'serial' (no collective shares the CUs with backward) stays unless another mode's median is more than 2 % faster."""
    from vnet_tensorflow_amd import parallel

    def run(costs):
        now = [0.0]
        t = parallel.StepModeAutotune(["segmented", "serial", "off"], steps=2, blocks=3, clock=lambda: now[0])
        while t.choice is None:
            m = t.mode()
            t.before()
            now[0] += costs[m]
            t.after()
        return t.choice, t.times
    assert run({"segmented": 0.0295, "serial": 0.0300, "off": 0.0400})[0] == "serial"       # 1.7 % faster: not enough
    assert run({"segmented": 0.0290, "serial": 0.0300, "off": 0.0400})[0] == "segmented"    # 3.3 % faster
    assert run({"segmented": 0.0310, "serial": 0.0300, "off": 0.0200})[0] == "off"
    choice, times = run({"segmented": 0.031, "serial": 0.030, "off": 0.040})
    assert choice == "serial" and abs(times[1] - 0.030) < 1e-12


def test_step_mode_autotune_single_candidate_is_free():
    from vnet_tensorflow_amd import parallel
    t = parallel.StepModeAutotune(["segmented"])
    assert t.choice == "segmented" and t.mode() == "segmented"
    t.before(); t.after()
    assert t.times == []


# ---- two-pass exchange: prefix buckets first (asynchronously), the rest after the second backward pass ---------------------
def _two_pass_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vnet_tensorflow_amd import optim, parallel
    parallel.init_from_env("gloo")
    params = _chain_model(5 + rank)
    flat = optim.FlatParams(params)          # gradient-production order: e, d, c, b, a
    parallel.broadcast_parameters(flat.data)
    sync = parallel.BucketedGradAllReduce(flat, bucket_bytes=256, phase1_last=1)      # pass 1 ends with d
    assert any(last == 2 for _, _, _, last in sync.buckets)
    sync.hold_all = True                      # segmented-graph mode: hooks only count
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(rank))
    flat.zero_grad()
    _chain_loss(params, x).backward()
    local = flat.grad.clone()
    sync.reduce_prefix()
    launched_prefix = list(sync._launched)
    sync.reduce_rest()
    torch.save({"local": local, "reduced": flat.grad.clone(), "prefix": launched_prefix,
                "buckets": [(f, l) for _, _, f, l in sync.buckets]}, os.path.join(out, "p%d.pt" % rank))
    dist.destroy_process_group()


def test_prefix_then_rest_reduction(tmp_path):
    mp.spawn(_two_pass_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / ("p%d.pt" % k)) for k in range(2)]
    assert torch.allclose(r[0]["reduced"], r[0]["local"] + r[1]["local"], atol=1e-6)
    assert torch.equal(r[0]["reduced"], r[1]["reduced"])
    for (first, last), went in zip(r[0]["buckets"], r[0]["prefix"]):
        assert went == (last <= 2), (first, last, went)          # exactly the buckets of pass 1 (variables e, d) left early
    assert any(r[0]["prefix"]) and not all(r[0]["prefix"])


# ---- world 8 on the REAL variable layout ----------------------------------------------------------------
def _world8_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from vnet_tensorflow_amd import networks, optim, parallel
    parallel.init_from_env("gloo")
    np.random.seed(1)
    net = networks.VNet(2, 0.0, 16, 4, (1, 2, 3, 3), 3, True, "prelu", device=torch.device("cpu"))
    net.build((1, 16, 16, 16, 1))
    flat = optim.FlatParams(net.named_parameters())
    names = flat.names
    enc = ("vnet/encoder", "vnet/input_layer")
    head = [i for i, n in enumerate(names) if not n.startswith(enc)]                 # model.image2label._setup_training
    assert head == list(range(len(head))) and len(head) < len(names)
    sync = parallel.BucketedGradAllReduce(flat, hold_fraction=0.0, phase1_last=head[-1])
    sync.hold_all = True
    k = sync.phase1_last
    p1 = [bi for bi, (_, _, _, last) in enumerate(sync.buckets) if last <= k + 1]
    frac = sum(4 * (sync.buckets[bi][1] - sync.buckets[bi][0]) for bi in p1) / (4.0 * flat.numel)
    # a "gradient" that tells rank and position apart, two-pass exchange as the segmented step graph runs it
    flat.grad.copy_(torch.arange(flat.numel, dtype=torch.float32).remainder_(97.0).add_(float(rank)))
    sync.reduce_prefix()
    launched_p1 = [bi for bi, _ in sync.launch_log]
    sync.reduce_rest()
    expect = torch.arange(flat.numel, dtype=torch.float32).remainder_(97.0).mul_(world).add_(float(sum(range(world))))
    ok = bool(torch.equal(flat.grad, expect))
    # every rank must leave the start-up autotune with the same mode (block times are max-reduced over the ranks)
    import time
    cost = {"segmented": 0.004 + 0.001 * (rank % 3), "serial": 0.004, "off": 0.003 + 0.004 * (rank == 5)}
    tuner = parallel.StepModeAutotune(["segmented", "serial", "off"], steps=2, blocks=3)
    while tuner.choice is None:
        m = tuner.mode()
        tuner.before()
        time.sleep(cost[m])
        tuner.after()
    torch.save({"frac": frac, "p1": p1, "launched_p1": launched_p1, "nb": len(sync.buckets), "ok": ok, "choice": tuner.choice,
                "times": tuner.times, "numel": flat.numel}, os.path.join(out, "w%d.pt" % rank))
    dist.destroy_process_group()


def test_world8_two_pass_exchange_on_the_full_width_layout(tmp_path):
    """8 ranks (gloo, CPU) on the full-width V-Net's 43.9 M-parameter flat layout: the buckets of backward pass 1 (output layer,
    decoder, bottom level) hold ~81 % of the gradient bytes and are exactly what reduce_prefix() sends while pass 2
    would run; after reduce_rest() every rank holds the sum; and all eight ranks leave the autotune with the same mode."""
    world = 8
    mp.spawn(_world8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / ("w%d.pt" % r)) for r in range(world)]
    assert all(r["ok"] for r in rs)
    assert all(r["numel"] >= 43940486 for r in rs)
    assert all(0.78 < r["frac"] < 0.84 for r in rs), rs[0]["frac"]
    assert all(r["launched_p1"] == r["p1"] and 0 < len(r["p1"]) < r["nb"] for r in rs)
    assert len({r["choice"] for r in rs}) == 1 and all(r["times"] == rs[0]["times"] for r in rs)
    assert rs[0]["choice"] == "serial"            # 'off' is slow on rank 5, 'segmented' on some ranks: max over ranks decides


# ---- round 6: bf16 gradient buckets with fp32 accumulation, and a step-length-aware preferred mode --------------
def _bf16_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from vnet_tensorflow_amd import optim, parallel
    parallel.init_from_env("gloo")
    g = torch.Generator().manual_seed(5)
    shapes = [("a/weights", (37, 11)), ("a/biases", (11,)), ("b/weights", (11, 129)), ("b/biases", (129,)), ("c/weights", (1000,))]
    params = [(n, torch.nn.Parameter(torch.randn(s, generator=g))) for n, s in shapes]
    res = {}
    for dt in ("fp32", "bf16"):
        flat = optim.FlatParams(params)
        sync = parallel.BucketedGradAllReduce(flat, bucket_bytes=1500, comm_dtype=dt)       # several buckets, sizes not multiples of world
        sync.hold_all = True
        gr = torch.Generator().manual_seed(1000 + rank)
        # gradients of very different magnitudes per variable, as a network's are
        scale = torch.ones(flat.numel)
        for i, (off, p) in enumerate(zip(flat.offsets, flat.params)):
            scale[off:off + p.numel()] = 10.0 ** (i - 2)
        flat.grad.copy_(torch.randn(flat.numel, generator=gr) * scale)
        mine = flat.grad.clone()
        sync.begin_step()
        sync.reduce_all()
        res[dt] = (flat.grad.clone(), sync.comm_bytes, sync.exposed_seconds())
        res["mine"] = mine
        sync.remove()
    torch.save(res, os.path.join(out, "b%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bf16_gradient_buckets_accumulate_in_fp32(tmp_path, world):
    """GradCommDtype 'bf16': bf16 on the links, fp32 accumulation on receipt.  Every rank ends with the SAME bits; the result is
    RNE_bf16(sum_r RNE_bf16(g_r)) with the sum taken in fp32 in rank order -- computed here independently; against the fp32
    all-reduce the difference is two roundings (bf16's unit roundoff is 2^-8: <= 2^-7 of sum_r |g_r| per element, far inside it in the norm); half the link bytes."""
    mp.spawn(_bf16_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / ("b%d.pt" % r)) for r in range(world)]
    for r in rs[1:]:
        assert torch.equal(r["bf16"][0], rs[0]["bf16"][0]) and torch.equal(r["fp32"][0], rs[0]["fp32"][0])
    # (the "mine" stored last is the bf16 leg's input = the fp32 leg's: same seed)
    parts = [r["mine"] for r in rs]
    acc = parts[0].to(torch.bfloat16).to(torch.float32)
    for p in parts[1:]:
        acc = acc + p.to(torch.bfloat16).to(torch.float32)
    want = acc.to(torch.bfloat16).to(torch.float32)
    got, ref = rs[0]["bf16"][0], rs[0]["fp32"][0]
    assert torch.equal(got, want)
    mag = sum(p.abs() for p in parts)
    assert bool(((got - ref).abs() <= 2.0 ** -7 * mag + 1e-30).all())
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    assert 0.45 < rs[0]["bf16"][1] / rs[0]["fp32"][1] < 0.56                    # half the bytes (+ padding to a multiple of world)
    assert rs[0]["bf16"][2] is not None and rs[0]["bf16"][2] > 0.0


def test_bf16_gradient_exchange_is_opt_in_and_needs_the_bf16_mode(tmp_path):
    from vnet_tensorflow_amd.model import image2label
    from tests.test_host import _config
    cfg = _config(tmp_path)
    m = image2label(None, cfg, device="cpu", verbose=False)
    m.read_config()
    assert m.grad_comm_dtype == "fp32"
    cfg["TrainingSetting"]["GradCommDtype"] = "bf16"
    with pytest.raises(SystemExit, match="ComputeDtype"):
        image2label(None, cfg, device="cpu", verbose=False).read_config()
    cfg["TrainingSetting"]["ComputeDtype"] = "bf16"
    cfg["TrainingSetting"]["Networks"]["NumChannel"] = 8
    m = image2label(None, cfg, device="cpu", verbose=False)
    m.read_config()
    assert m.grad_comm_dtype == "bf16"


def test_step_mode_autotune_preference_follows_the_exposed_all_reduce():
    """'serial' is the default only while its exposed all-reduce is a small share of the step (<= 5 %).  A 25 ms fp32 step with
    1 ms of all-reduce keeps it (segmented 1.7 % faster: inside the margin); a 5.3 ms bf16 step with the same 1 ms exposed (19 %)
    prefers the overlapped replay even when it measures the SAME or up to 2 % slower -- and still takes 'serial' when serial is
    more than 2 % faster than every overlapped mode."""
    from vnet_tensorflow_amd import parallel

    def run(costs, exposed):
        now = [0.0]
        t = parallel.StepModeAutotune(["segmented", "serial", "off"], steps=2, blocks=3, clock=lambda: now[0], exposed=lambda: exposed)
        while t.choice is None:
            m = t.mode()
            t.before()
            now[0] += costs[m]
            t.after()
        return t
    t = run({"segmented": 0.02458, "serial": 0.0250, "off": 0.0400}, 0.0010)
    assert t.choice == "serial" and t.preferred == "serial" and abs(t.exposed_fraction - 0.04) < 1e-9
    t = run({"segmented": 0.00535, "serial": 0.0053, "off": 0.0090}, 0.0010)
    assert t.choice == "segmented" and t.preferred == "segmented" and t.exposed_fraction > 0.18
    t = run({"segmented": 0.0060, "serial": 0.0053, "off": 0.0090}, 0.0010)          # overlap hurts the kernels more than it hides
    assert t.choice == "serial" and t.preferred == "segmented"
    t = run({"segmented": 0.0060, "serial": 0.0059, "off": 0.0052}, 0.0010)          # the eager enqueue is the fastest overlapped mode
    assert t.choice == "off" and t.preferred == "off"
    t = run({"segmented": 0.0250, "serial": 0.0250, "off": 0.0400}, None)             # no measurement: round 5's behaviour
    assert t.choice == "serial" and t.exposed_fraction == 0.0

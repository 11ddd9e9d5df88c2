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

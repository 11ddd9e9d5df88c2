"""EMULATION (one GPU), not a scaling measurement: what does a resident collective cost the data-parallel step in each step mode?

The training step runs its real data-parallel code path in a process group of ONE rank (VNET_DP_FORCE=1); every bucket
all-reduce is replaced by a spinner kernel shaped like RCCL's gfx950 all-reduce kernel (profiles/probes/rccl_like.hip: 256-thread
workgroups, 280 registers, 19.7 KB LDS, one per channel) that stays resident on the communication stream for
bucket_bytes / total_bytes x ALLREDUCE_MS.  It holds CU slots the way the collective does; it moves no data, so HBM and xGMI
contention are not modelled.  Modes: 'segmented' (pass-1 buckets under the encoder's backward, two-pass graphs), 'serial' (same
graphs, every all-reduce after backward: fully exposed, nothing shares the chip), 'off' (eager enqueue, buckets from the hooks).

    python profiles/dp_emulation.py [fp32|bf16]      -> table of ms/step per (mode, channels, all-reduce ms)
"""
import ctypes
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ["VNET_DP_FORCE"] = "1"
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M, ops, parallel
from vnet_tensorflow_amd.data import synthetic_case

compute = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cin, K = (4, 5) if compute == "bf16" else (1, 2)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=0, world_size=1)
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "librccl_like.so"))
L.rccl_like.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
EMU = {"channels": 0, "ms": 0.0, "total": 1.0}


class _Done(object):
    def wait(self):
        return True


def fake_all_reduce(view, op=None, group=None, async_op=False):
    if EMU["channels"] > 0 and EMU["ms"] > 0:
        ms = EMU["ms"] * view.numel() * 4.0 / EMU["total"]
        L.rccl_like(EMU["channels"], ms, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return _Done()


parallel.dist.all_reduce = fake_all_reduce          # (parallel.py calls dist.all_reduce for every bucket)
im, lb = synthetic_case([128] * 3, cin, K, 1000)
images = torch.from_numpy(im[None]).to(dev)
labels = torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)
print("# EMULATION on one GPU (spinner instead of RCCL; no data moves): %s net, 128^3, batch 1" % compute)
print("# %-10s %8s %8s %10s" % ("mode", "channels", "allreduce_ms", "ms/step"))
for mode in ("serial", "segmented", "off"):
    os.environ["VNET_DP_MODE"] = mode
    os.environ["VNET_STEP_GRAPH"] = "0" if mode == "off" else "1"
    np.random.seed(42)
    ops.clear_pack_registry()
    m = M.image2label(None, bench.config(128, 1, cin, K, compute), device=dev, verbose=False)
    m.rank, m.local_rank, m.world = 0, 0, 1
    m.read_config(); m.build_model_graph(); m._setup_training()
    EMU["total"] = float(m.flat.numel * 4)
    EMU["channels"], EMU["ms"] = 0, 0.0
    for _ in range(6):
        m.train_step(images, labels)
    torch.cuda.synchronize()
    assert m.step_mode() == mode, (m.step_mode(), mode)
    for channels, ms in ((0, 0.0), (16, 1.2), (32, 1.2), (16, 2.5), (32, 2.5)):
        EMU["channels"], EMU["ms"] = channels, ms
        for _ in range(3):
            m.train_step(images, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            m.train_step(images, labels)
        torch.cuda.synchronize()
        print("  %-10s %8d %8.1f %10.3f" % (mode, channels, ms, (time.perf_counter() - t0) / n * 1e3), flush=True)
    del m
    ops.set_compute_dtype("fp32")
dist.destroy_process_group()

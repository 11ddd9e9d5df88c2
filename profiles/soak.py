"""Soak: N replayed training steps at 128^3 over four alternating batches (default 3000, ~80 s), memory and loss sampled on the
way -- the replayed step graph must neither leak nor stall.  Round 2 on one MI355X, 8000 steps each: fp32 allocated 1851.6 MB /
reserved 3736.0 MB at steps 10, 800, 4000 and 7999 (~680 MB of it the per-layer filter-gradient slabs of the batched reduce), loss
0.558 -> 0.00007, 25.43 ms/step sustained; bf16 (1 modality) 1642.1 / 4032.0 MB, 7.24 ms/step.
    python profiles/soak.py [steps] [fp32|bf16]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M, ops
from vnet_tensorflow_amd.data import synthetic_case

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
compute = sys.argv[2] if len(sys.argv) > 2 else "fp32"
P = 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
np.random.seed(42)
m = M.image2label(None, bench.config(P, 1, 1, 2, compute), device=dev, verbose=False)
m.read_config(); m.build_model_graph(); m._setup_training()
batches = []
for s in range(4):
    im, lb = synthetic_case([P] * 3, 1, 2, 1000 + s)
    batches.append((torch.from_numpy(im[None]).to(dev), torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)))
t0 = time.perf_counter()
marks = sorted(set([10, steps // 10, steps // 2, steps - 1]))
for i in range(steps):
    loss = m.train_step(*batches[i % 4])
    if i in marks:
        torch.cuda.synchronize()
        print("step %5d  loss %.5f  alloc %.1f MB  reserved %.1f MB  %.2f ms/step so far" % (
            i, float(loss), torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20,
            (time.perf_counter() - t0) / (i + 1) * 1e3), flush=True)
assert np.isfinite(float(loss))
print("mode", m.step_mode(), "ok")

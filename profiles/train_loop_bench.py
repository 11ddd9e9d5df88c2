"""Throughput of the PRODUCT training loop, `image2label.train()` (reference model.py:632-815), on Data.Synthetic at the
bench's patch size -- PCIe-inclusive: every step's batch is cropped from a cached host volume by the prefetch threads,
staged in pinned memory, copied host-to-device on the copy stream and fed to the replayed step graph; the loss of step
t-1 is read while step t runs.  Compare with `python bench.py` (inputs resident in HBM).
    python profiles/train_loop_bench.py [patch] [fp32|bf16] [channels] [classes]"""
import json
import os
import sys
import tempfile
import numpy as np
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M

P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
compute = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 1
K = int(sys.argv[4]) if len(sys.argv) > 4 else 2
tmp = tempfile.mkdtemp()
cfg = bench.config(P, 1, cin, K, compute)
T = cfg["TrainingSetting"]
T["Data"]["Synthetic"] = {"Cases": 8, "Shape": [P + 16] * 3}        # random 128^3 crops of 144^3 volumes
T.update(Epoches=9, LogInterval=10 ** 9, LogDir=os.path.join(tmp, "log"), CheckpointDir=os.path.join(tmp, "ckpt"), Testing=False)
np.random.seed(42)
m = M.image2label(None, cfg, verbose=False)
m.train()
out = {"what": "image2label.train() steady state, PCIe-inclusive", "patch": P, "compute": compute, "channels": cin, "classes": K,
       "steps_timed": m.steps_timed, "patches_per_s": round(m.steps_timed / m.seconds_timed, 3),
       "ms_per_step": round(m.seconds_timed / m.steps_timed * 1e3, 3), "step_enqueue": m.step_mode(), "final_loss": m.last_loss}
print(json.dumps(out))

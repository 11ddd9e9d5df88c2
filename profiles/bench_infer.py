"""Sliding-window inference throughput of the native driver (csrc/vnet_infer.cpp) on a synthetic 256^3 volume:
full-width V-Net, 128^3 patches, stride 64, batch 2 -- fp32 and bf16 compute.   python profiles/bench_infer.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import time
dev = torch.device("cuda", 0)
np.random.seed(42)
cfg = bench.config(128, 2, 1, 2, "fp32")
cfg["EvaluationSetting"] = {"Stride": [64, 64, 64], "BatchSize": 2, "ProbabilityOutput": False}
m = M.image2label(None, cfg, device=dev, verbose=False)
m.read_config(); m.build_model_graph()
with tempfile.TemporaryDirectory() as tmp:
    w, v = os.path.join(tmp, "net.vnetw"), os.path.join(tmp, "vol.npy")
    M.export_weights(m.network, w)
    rng = np.random.default_rng(0)
    vol = np.clip(127.5 + 40 * rng.standard_normal((256, 256, 256, 1)), 0, 255).astype(np.float32)
    np.save(v, vol)
    # the Python evaluate path (image2label.evaluate_single_3D): 27 patches + the duplicated last batch = 14 batches of 2
    from vnet_tensorflow_amd import ops
    for compute in ("fp32", "fp32_split3", "bf16"):
        with ops.context(m.ctx):                                  # (round 4: the compute dtype is per-model state)
            ops.set_compute_dtype(compute)
        m.evaluate_single_3D(vol[:192, :192, :192])               # warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.evaluate_single_3D(vol)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("python evaluate_single_3D %s: %.3f s = %.1f patches/s (crop + H2D + forward + accumulate + argmax + D2H)" % (compute, dt, 28 / dt))
    ops.set_compute_dtype("fp32")
    del m
    torch.cuda.empty_cache()
    for compute in ("fp32", "fp32_split3", "bf16"):
        out = subprocess.run([os.path.join(ROOT, "vnet_tensorflow_amd", "vnet_infer"), "--weights", w, "--image", v,
                              "--label-out", os.path.join(tmp, "lab.npy"), "--classes", "2", "--channels", "16", "--levels", "4",
                              "--convs", "1,2,3,3", "--bottom", "3", "--patch", "128,128,128", "--stride", "64,64,64", "--batch", "2",
                              "--compute", compute], capture_output=True, text=True, timeout=900)
        print(out.stdout.strip() or out.stderr[-500:])

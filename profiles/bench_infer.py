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


class A:
    pass


args = A(); args.channels = 1; args.classes = 2; args.batch = 2; args.patch = 128; args.compute = "fp32"
dev = torch.device("cuda", 0)
np.random.seed(42)
m = M.image2label(None, bench.config(args), device=dev, verbose=False)
m.read_config(); m.build_model_graph()
with tempfile.TemporaryDirectory() as tmp:
    w, v = os.path.join(tmp, "net.vnetw"), os.path.join(tmp, "vol.npy")
    M.export_weights(m.network, w)
    rng = np.random.default_rng(0)
    np.save(v, np.clip(127.5 + 40 * rng.standard_normal((256, 256, 256, 1)), 0, 255).astype(np.float32))
    del m
    torch.cuda.empty_cache()
    for compute in ("fp32", "bf16"):
        out = subprocess.run([os.path.join(ROOT, "vnet_tensorflow_amd", "vnet_infer"), "--weights", w, "--image", v,
                              "--label-out", os.path.join(tmp, "lab.npy"), "--classes", "2", "--channels", "16", "--levels", "4",
                              "--convs", "1,2,3,3", "--bottom", "3", "--patch", "128,128,128", "--stride", "64,64,64", "--batch", "2",
                              "--compute", compute], capture_output=True, text=True, timeout=900)
        print(out.stdout.strip() or out.stderr[-500:])

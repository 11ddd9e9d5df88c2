"""How sensitive is the training step to losing compute units (a concurrent RCCL collective holds some)?
   python profiles/cu_steal_bench.py [--compute bf16]   -> ms/step with 0 / 16 / 32 / 64 CUs held by a spinning kernel."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M
from vnet_tensorflow_amd.data import synthetic_case


class A:
    pass


args = A(); args.channels = 1; args.classes = 2; args.batch = 1; args.patch = 128
args.compute = "bf16" if "bf16" in sys.argv else "fp32"
dev = torch.device("cuda", 0)
np.random.seed(42)
m = M.image2label(None, bench.config(args), device=dev, verbose=False)
m.rank, m.local_rank, m.world = 0, 0, 1
m.read_config(); m.build_model_graph(); m._setup_training()
im, lb = synthetic_case([128] * 3, 1, 2, 1000)
images = torch.from_numpy(im[None]).to(dev); labels = torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libcu_steal.so"))
L.cu_steal.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
side = torch.cuda.Stream()
for _ in range(3):
    m.train_step(images, labels)
torch.cuda.synchronize()
from vnet_tensorflow_amd import ops
TABLE = "table" in sys.argv
for held in ((0, 16) if TABLE else (0, 16, 32, 64)):
    steps = 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if held:
        L.cu_steal(held, 400.0, ctypes.c_void_p(side.cuda_stream))     # outlives the timed steps
        time.sleep(0.002)
    if TABLE:
        ops.profile_start()
    for _ in range(steps):
        m.train_step(images, labels)
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    if TABLE:
        torch.cuda.current_stream().synchronize()
        ops._PROFILE["on"] = False
        per = {}
        for (t, f, b, e0, e1) in ops._PROFILE["records"]:
            per[t] = per.get(t, 0.0) + e0.elapsed_time(e1) / steps
        ops._PROFILE["records"] = []
        if held == 0:
            base = per
        else:
            for t, v in sorted(per.items(), key=lambda kv: -(kv[1] - base.get(kv[0], 0)))[:16]:
                print("   %-40s %7.3f -> %7.3f ms/step  x%.2f" % (t, base.get(t, 0), v, v / max(base.get(t, 1e-9), 1e-9)))
    torch.cuda.synchronize()
    print("CUs held %3d : %.2f ms/step  (ideal %.2f)" % (held, dt, 0), flush=True)

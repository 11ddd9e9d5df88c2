"""N training steps of the bench workload and nothing else (for rocprofv3 --kernel-trace timelines of graph replay vs
eager enqueue):  python profiles/step_only.py [steps] [fp32|bf16] [channels] [classes]     (VNET_STEP_GRAPH=0|1)"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M
from vnet_tensorflow_amd.data import synthetic_case

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
compute = sys.argv[2] if len(sys.argv) > 2 else "fp32"
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 1
K = int(sys.argv[4]) if len(sys.argv) > 4 else 2
P = 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
np.random.seed(42)
m = M.image2label(None, bench.config(P, 1, cin, K, compute), device=dev, verbose=False)
m.read_config(); m.build_model_graph(); m._setup_training()
im, lb = synthetic_case([P] * 3, cin, K, 1000)
images = torch.from_numpy(im[None]).to(dev); labels = torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)
for _ in range(4):
    m.train_step(images, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m.train_step(images, labels)
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("mode %s  %.3f ms/step  host enqueue %.3f ms/step" % (m._graph_mode(), dt / steps * 1e3, th / steps * 1e3))

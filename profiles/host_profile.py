"""Host-side cost of enqueueing one training step (cProfile over N steps; the GPU runs behind).
   python profiles/host_profile.py [fp32|bf16] [patch]"""
import cProfile, pstats, sys, io
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M
from vnet_tensorflow_amd.data import synthetic_case

class A: pass
args = A(); args.channels = 1; args.classes = 2; args.batch = 1
args.compute = sys.argv[1] if len(sys.argv) > 1 else "fp32"
args.patch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
np.random.seed(42)
m = M.image2label(None, bench.config(args), device=dev, verbose=False)
m.rank, m.local_rank, m.world = 0, 0, 1
m.read_config(); m.build_model_graph(); m._setup_training()
im, lb = synthetic_case([args.patch] * 3, 1, 2, 1000)
images = torch.from_numpy(im[None]).to(dev); labels = torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)
for _ in range(3):
    m.train_step(images, labels)
torch.cuda.synchronize()
N = 10
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    m.train_step(images, labels)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
print(s.getvalue().replace("/root/repo/", "")[:6000])

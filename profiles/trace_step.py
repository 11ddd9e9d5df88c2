"""One training step under torch.profiler: which host ops launch the small copy / elementwise kernels?"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M
from vnet_tensorflow_amd.data import synthetic_case

class A: pass
args = A(); args.channels = 1; args.classes = 2; args.batch = 1; args.patch = int(sys.argv[1]) if len(sys.argv) > 1 else 64; args.compute = "fp32"
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.train_step(images, labels)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
for name in ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::sum"):
    evs = [e for e in prof.events() if e.name == name]
    print("==", name, len(evs))
    seen = {}
    for e in evs:
        st = tuple(s for s in (e.stack or []) if 'vnet_tensorflow_amd' in s or 'bench' in s)[:3]
        seen[st] = seen.get(st, 0) + 1
    for st, n in sorted(seen.items(), key=lambda kv: -kv[1])[:8]:
        print("   ", n, " <- ".join(s.split('/')[-1] for s in st))

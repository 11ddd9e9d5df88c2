import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M, ops
from vnet_tensorflow_amd.data import synthetic_case
class A: pass
args = A(); args.channels = 1; args.classes = 2; args.batch = 1; args.compute = "fp32"; args.patch = 96
dev = torch.device("cuda", 0); np.random.seed(42)
m = M.image2label(None, bench.config(args), device=dev, verbose=False)
m.rank, m.local_rank, m.world = 0, 0, 1
m.read_config(); m.build_model_graph(); m._setup_training()
im, lb = synthetic_case([96] * 3, 1, 2, 1000)
x = torch.from_numpy(im[None]).to(dev); y = torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)
for i in range(2001):
    loss = m.train_step(x, y)
    if i in (50, 500, 1000, 2000):
        torch.cuda.synchronize()
        print(i, "loss %.5f" % float(loss), "alloc %.1f MB reserved %.1f MB keep %d retired %d" % (torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, len(ops._PG["keep"]), len(ops._WS_RETIRED)))

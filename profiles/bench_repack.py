"""Time of the one-launch filter repack (ops.repack_registered: every registered filter image of the network, after each optimiser
step) for the bench network:  python profiles/bench_repack.py [fp32|fp32_split3|bf16] [reps]     (VNET_PACK_BOTH_X3=0: the f32x3 images
one at a time, the round-5 form)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from vnet_tensorflow_amd import model as M, ops
from vnet_tensorflow_amd.data import synthetic_case

compute = sys.argv[1] if len(sys.argv) > 1 else "fp32_split3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
P = 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
np.random.seed(42)
cin, K = (4, 5) if compute == "bf16" else (1, 2)
m = M.image2label(None, bench.config(P, 1, cin, K, compute), device=dev, verbose=False)
m.read_config(); m.build_model_graph(); m._setup_training()
im, lb = synthetic_case([P] * 3, cin, K, 1000)
images = torch.from_numpy(im[None]).to(dev); labels = torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)
for _ in range(3):
    m.train_step(images, labels)
torch.cuda.synchronize()
with ops.context(m.ctx):
    for _ in range(3):
        ops.repack_registered()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.repack_registered()
    e1.record()
    torch.cuda.synchronize()
    ents = [(w(), key) for w, key, _ in ops._PACK_REG["entries"]]
nelem = sum(int(np.prod(w.shape)) for w, _ in ents if w is not None)
print("%s: repack of %d registered images (%.1f M filter elements, each counted once per image) %.1f us" % (compute, len(ents), nelem / 1e6, e0.elapsed_time(e1) / reps * 1e3))

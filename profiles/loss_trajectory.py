"""Step-by-step loss of the SAME training run (same seed, same four alternating 128^3 batches) in several compute modes: how fast the
trajectories separate.  fp32 twice (determinism), fp32_split3, bf16.   python profiles/loss_trajectory.py [steps]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from vnet_tensorflow_amd import model as M
from vnet_tensorflow_amd.data import synthetic_case

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
P = 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
batches = []
for s in range(4):
    im, lb = synthetic_case([P] * 3, 1, 2, 1000 + s)
    batches.append((torch.from_numpy(im[None]).to(dev), torch.from_numpy(lb[None, ..., None].astype(np.int32)).to(dev)))
out = {}
for tag, compute in (("fp32", "fp32"), ("fp32 again", "fp32"), ("fp32_split3", "fp32_split3"), ("bf16", "bf16")):
    np.random.seed(42)
    m = M.image2label(None, bench.config(P, 1, 1, 2, compute), device=dev, verbose=False)
    m.read_config(); m.build_model_graph(); m._setup_training()
    ls = []
    for i in range(steps):
        ls.append(float(m.train_step(*batches[i % 4])))
    out[tag] = np.array(ls)
    del m
    torch.cuda.empty_cache()
ref = out["fp32"]
print("step      fp32 loss   |fp32 again - fp32|   |split3 - fp32|   |bf16 - fp32|   (relative to fp32)")
for i in list(range(0, 12)) + list(range(12, steps, max(1, steps // 20))):
    print("%4d  %12.8f   %.2e            %.2e        %.2e" % (i, ref[i], abs(out["fp32 again"][i] - ref[i]) / ref[i],
                                                              abs(out["fp32_split3"][i] - ref[i]) / ref[i], abs(out["bf16"][i] - ref[i]) / ref[i]))

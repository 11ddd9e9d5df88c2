"""Robustness sweep: one training step (fwd + loss + bwd + optimiser) over odd shapes / batch sizes / channel counts /
both network variants / both arithmetic modes / several losses and optimisers.  Prints ms and the loss; any exception
or non-finite loss is a failure.   python profiles/sweep_configs.py"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import model as M, ops
from vnet_tensorflow_amd.data import synthetic_case

dev = torch.device("cuda", 0)
cases = [
    # patch, batch, cin, K, net, nch, levels, convs, bottom, compute, loss, opt
    ((128, 128, 128), 2, 1, 2, "VNet", 16, 4, [1, 2, 3, 3], 3, "fp32", "sorensen", "Adam"),
    ((96, 96, 96), 1, 1, 2, "VNet", 16, 4, [1, 2, 3, 3], 3, "fp32", "jaccard", "Adam"),
    ((80, 96, 112), 1, 4, 5, "VNet", 16, 4, [1, 2, 3, 3], 3, "fp32", "weighted_sorensen", "Momentum"),
    ((80, 96, 112), 1, 4, 5, "VNet", 16, 4, [1, 2, 3, 3], 3, "bf16", "mixed_sorensen", "Adam"),
    ((72, 88, 104), 1, 2, 3, "VNet", 16, 3, [1, 2, 3], 2, "fp32", "xent", "SGD"),
    ((36, 44, 52), 2, 1, 2, "VNet", 8, 2, [2, 2], 1, "fp32", "sorensen", "Adam"),
    ((64, 64, 64), 3, 1, 2, "VNet", 16, 4, [1, 2, 3, 3], 3, "bf16", "sorensen", "Adam"),
    ((100, 60, 44), 2, 3, 4, "VNet", 16, 2, [1, 2], 2, "fp32", "mixed_weighted_jaccard", "Adam"),
    ((128, 128, 128), 1, 1, 2, "VNet", 32, 4, [1, 2, 3, 3], 3, "fp32", "sorensen", "Adam"),
    ((160, 160, 160), 1, 1, 2, "VNet", 16, 4, [1, 2, 3, 3], 3, "fp32", "sorensen", "Adam"),
    ((192, 192, 192), 1, 4, 5, "VNet", 16, 4, [1, 2, 3, 3], 3, "bf16", "sorensen", "Adam"),
    ((60, 52, 44), 2, 1, 2, "VNet", 6, 3, [1, 2, 2], 2, "fp32", "sorensen", "Adam"),          # 6/12/24/48 channels
    ((45, 51, 39), 1, 3, 4, "VNet", 10, 2, [2, 1], 1, "bf16", "weighted_sorensen", "Adam"),   # odd extents, 10/20 channels
]
ok = True
for (P, B, cin, K, net, nch, lev, convs, bot, comp, loss, opt) in cases:
    cfg = {"TrainingSetting": {
        "Data": {"TrainingDataDirectory": "synthetic", "TestingDataDirectory": "synthetic",
                 "ImageFilenames": ["i%d.nii" % i for i in range(cin)], "LabelFilename": "l.nii", "Synthetic": {"Cases": 1}},
        "SegmentationClasses": list(range(K)), "BatchSize": B, "PatchShape": list(P), "ComputeDtype": comp,
        "Networks": {"Name": net, "Dropout": 0.0, "NumChannel": nch, "NumLevels": lev, "NumConvolutions": convs, "BottomConvolutions": bot},
        "Optimizer": {"Name": opt, "InitialLearningRate": 1e-3, "Momentum": 0.9, "Decay": {"Factor": 0.99, "Steps": 100}},
        "Loss": {"Name": loss, "Weights": [1.0 / (k + 1) for k in range(K)], "Alpha": 0.5,
                 "AllowPlainXent": True}}}       # (the reference itself exits on "xent": model.py:495-559)
    tag = "%s B%d cin%d K%d %s nch%d L%d %s %s %s" % (P, B, cin, K, net, nch, lev, comp, loss, opt)
    try:
        np.random.seed(1)
        m = M.image2label(None, cfg, device=dev, verbose=False)
        m.rank, m.local_rank, m.world = 0, 0, 1
        m.read_config(); m.build_model_graph(); m._setup_training()
        ims, lbs = [], []
        for b in range(B):
            im, lb = synthetic_case(list(P), cin, K, 77 + b)
            ims.append(im); lbs.append(lb[..., None])
        x = torch.from_numpy(np.stack(ims)).to(dev); y = torch.from_numpy(np.stack(lbs).astype(np.int32)).to(dev)
        l0 = float(m.train_step(x, y))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        l1 = float(m.train_step(x, y)); l2 = float(m.train_step(x, y))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2 * 1e3
        good = all(np.isfinite(v) for v in (l0, l1, l2))
        ok &= good
        print("%-90s %8.2f ms/step  loss %.4f %.4f %.4f  %s  mem %.1f GB" % (tag, dt, l0, l1, l2, "ok" if good else "NON-FINITE",
                                                                        torch.cuda.max_memory_allocated() / 2**30), flush=True)
        del m, x, y
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    except Exception as e:                                              # noqa: BLE001
        ok = False
        print("%-90s FAILED: %s: %s" % (tag, type(e).__name__, str(e)[:300]), flush=True)
    finally:
        ops.set_compute_dtype("fp32")
sys.exit(0 if ok else 1)

"""Seed spread of the whole-network gradient error (VERDICT r5 next #1a): the full-width V-Net's training step at the bench sizes against
the fp64-oracle fixtures of FIVE (weight seed, input seed) draws per config -- tests/golden/{c3_128cube, c2_64cube_b2}.npz (draw 0) and
tests/golden/spread/*_s{1..4}.npz -- in both fp32 modes.  Per run: worst / median sampled rel-L2 over the gradient tensors, worst head
error, whole-vector rel-L2; then max-over-seeds per (config, mode) and the ratio fp32_split3 / fp32 that the promotion rule reads
(<= 1.25).  Usage (GPU box):  python profiles/golden_seed_spread.py > gpurun_out/r06_golden_seed_spread.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_hip_golden_full as T  # noqa: E402
from tests.golden.make_golden_full import CASES  # noqa: E402

dev = torch.device("cuda", 0)
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
table = {}
print("%-6s %-12s %10s %10s %10s %10s %10s %10s  %s" % ("case", "mode", "worst", "median", "head", "norm", "vector", "loss err", "worst tensor"))
for cfg in ("c3", "c2"):
    for s in range(10):
        case = cfg if s == 0 else "%ss%d" % (cfg, s)
        if not os.path.exists(os.path.join(GOLD, CASES[case][0])):
            print("%-6s (fixture missing)" % case)
            continue
        for mode in ("fp32", "fp32_split3"):
            z, net, logits, loss, sm, pred, lab, K = T._run_case(dev, case, mode)
            errs = T._grad_errors(z, net)
            names = list(map(str, z["names"]))
            num = sum((e[1] * float(z["grad_norm"][names.index(e[0])])) ** 2 for e in errs)
            den = sum(float(v) ** 2 for v in z["grad_norm"])
            w = max(errs, key=lambda e: e[1])
            row = (w[1], float(np.median([e[1] for e in errs])), max(e[3] for e in errs), max(e[2] for e in errs), (num / den) ** 0.5,
                   abs(loss - float(z["loss"])))
            table.setdefault((cfg, mode), []).append(row)
            print("%-6s %-12s %10.3e %10.3e %10.3e %10.3e %10.3e %10.3e  %s" % ((case, mode) + row + (w[0],)), flush=True)
            del net, logits, sm, pred
            torch.cuda.empty_cache()
print()
print("max over seeds (and mean of the per-seed worst):")
for cfg in ("c3", "c2"):
    for mode in ("fp32", "fp32_split3"):
        rows = np.array(table.get((cfg, mode), [[np.nan] * 6]))
        print("%-4s %-12s n=%d  worst %.3e (mean %.3e)  median %.3e  head %.3e  norm %.3e  vector %.3e" %
              (cfg, mode, len(rows), rows[:, 0].max(), rows[:, 0].mean(), rows[:, 1].max(), rows[:, 2].max(), rows[:, 3].max(), rows[:, 4].max()))
    a, b = np.array(table.get((cfg, "fp32_split3"), [[np.nan] * 6])), np.array(table.get((cfg, "fp32"), [[np.nan] * 6]))
    print("%-4s ratio fp32_split3 / fp32 of the max-over-seeds: worst %.3f  median %.3f  head %.3f  vector %.3f" %
          (cfg, a[:, 0].max() / b[:, 0].max(), a[:, 1].max() / b[:, 1].max(), a[:, 2].max() / b[:, 2].max(), a[:, 4].max() / b[:, 4].max()))

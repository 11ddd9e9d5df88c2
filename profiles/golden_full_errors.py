import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests import test_hip_golden_full as T
dev = torch.device("cuda", 0)
import os
compute = os.environ.get("GOLDEN_COMPUTE")          # e.g. fp32_split3 (default: the fixture's own mode)
for case in sys.argv[1:]:
    z, net, logits, loss, sm, pred, lab, K = T._run_case(dev, case, compute)
    errs = T._grad_errors(z, net)
    print(case, "loss err", abs(loss - float(z["loss"])))
    for e in sorted(errs, key=lambda e: -e[1])[:25]:
        print("%-70s sample %.2e norm %.2e head %.2e sum %.2e" % e)
    ws = [e[1] for e in errs if e[0].endswith("weights")]
    vs = [e[1] for e in errs if not e[0].endswith("weights")]
    print("weights: max %.2e median %.2e ; vectors: max %.2e median %.2e" % (max(ws), np.median(ws), max(vs), np.median(vs)))

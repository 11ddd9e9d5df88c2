"""How far does the fp64 oracle's OWN bf16-operand C5 step move when its input is perturbed at fp32 round-off level?
Operand rounding is discontinuous: a 1e-7 relative change flips the bf16 rounding of a few operands (each flip is a 2^-8
relative jump), and the batch-norm backward passes amplify what flows back.  This is the yardstick for
the C5 gradient comparison in tests/test_hip_golden_full.py: the HIP path cannot agree with the fixture better than the
oracle agrees with itself.      python profiles/oracle_bf16_sensitivity.py   (about 12 minutes on 8 cores)"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import vnet_oracle as O
from tests.golden.make_golden_full import CASES, sample_indices
from tests.util import rel_l2

fname, P, B, cin, K, seed, rounding = CASES["c5"]
z = np.load(os.path.join("tests", "golden", fname))
ps = O.ParamStore(rng=np.random.default_rng(42))
net = O.VNetOracle(K, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", ps)
x, lab = O.synthetic_batch(B, P, cin, K, seed=seed)
xp = x.astype(np.float64) * (1.0 + 1e-7 * np.random.default_rng(1).standard_normal(x.shape))
O.CONV5_OPERAND_ROUNDING = rounding
res = O.run_step(xp, lab, net, "sorensen")
O.CONV5_OPERAND_ROUNDING = None
print("oracle(bf16 operands) vs itself under a 1e-7 relative input perturbation: |dloss| = %.3e" % abs(res["loss"] - float(z["loss"])))
s = (slice(None),) + (slice(None, None, 4),) * 3
print("logits sample rel-L2 %.3e" % rel_l2(res["logits"][s], z["logits_sample"]))
errs = []
for i, n in enumerate(map(str, z["names"])):
    gn = float(z["grad_norm"][i])
    if gn < 1e-7:
        continue
    g = res["grads"][n].ravel()
    idx = sample_indices(i, g.size)
    errs.append((n, rel_l2(g[idx], z["grad_sample"][i][:len(idx)].astype(np.float64)), gn))
for e in sorted(errs, key=lambda e: -e[1])[:12]:
    print("%-70s sample %.2e" % e[:2])
ws = [e[1] for e in errs if e[0].endswith("weights")]
vs = [e[1] for e in errs if not e[0].endswith("weights")]
print("weights: max %.2e median %.2e ; vectors: max %.2e median %.2e" % (max(ws), np.median(ws), max(vs), np.median(vs)))
num = sum((e[1] * e[2]) ** 2 for e in errs)
den = sum(e[2] ** 2 for e in errs)
print("whole gradient vector: %.3e" % (num / den) ** 0.5)

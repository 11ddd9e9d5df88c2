import sys, numpy as np, torch
sys.path.insert(0, '.')
import tests.test_hip_golden_full as T
dev = torch.device('cuda', 0)
for case in ("c5", "c5s"):
    z, net, logits, loss, sm, pred, lab, K = T._run_case(dev, case)
    s = (slice(None),) + (slice(None, None, T.STRIDE),) * 3
    got, ref = logits[s].cpu().numpy(), z["logits_sample"]
    errs = T._grad_errors(z, net)
    names = list(map(str, z["names"]))
    num = sum((e[1] * float(z["grad_norm"][names.index(e[0])])) ** 2 for e in errs)
    den = sum(float(v) ** 2 for v in z["grad_norm"])
    print(case, "logits rel-L2 %.4e  loss err %.3e  pred agree %.5f  whole-gradient rel-L2 %.4f" % (
        T.rel_l2(got, ref), abs(loss - float(z["loss"])), (pred[s].cpu().numpy() == z["pred_sample"]).mean(), (num / den) ** 0.5), flush=True)

"""What does STOCK PyTorch-CPU fp32 (oneDNN convs, torch autograd; oracle/torch_ref.py) measure against the same fp64
fixtures?  Same recipe weights / inputs as tests/test_hip_golden_full.py, same per-tensor sample metric.  Calibrates which
part of the HIP path's gradient error is fp32 round-off of the problem itself (BN-coupled backward through ~80 layers).
    python profiles/golden_full_errors_cpu.py c2 [c3]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import torch_ref as T
from oracle import vnet_oracle as O
from tests.golden.make_golden_full import CASES, sample_indices
from tests.util import rel_l2

for case in sys.argv[1:]:
    fname, P, B, cin, K, seed, rounding = CASES[case]
    z = np.load(os.path.join("tests", "golden", fname))
    store = O.ParamStore(rng=np.random.default_rng(42))
    ref_net = O.VNetOracle(K, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", store)
    ref_net.GetNetwork(np.zeros((1, 16, 16, 16, cin)))
    params = {k: torch.tensor(v.v, dtype=torch.float32, requires_grad=True) for k, v in store.vars.items()}
    net = T.TorchVNet(K, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", params=params, dtype=torch.float32)
    x, lab = O.synthetic_batch(B, P, cin, K, seed=seed)
    loss, sm = T.loss_head(net.forward(torch.from_numpy(x)), torch.from_numpy(lab), "sorensen")
    loss.backward()
    print(case, "torch-cpu fp32 loss err %.2e" % abs(float(loss) - float(z["loss"])))
    errs = []
    for i, n in enumerate(map(str, z["names"])):
        gn = float(z["grad_norm"][i])
        g = params[n].grad
        if g is None or gn < 1e-7:
            continue
        got = g.numpy().astype(np.float64).ravel()
        idx = sample_indices(i, got.size)
        errs.append((n, rel_l2(got[idx], z["grad_sample"][i][:len(idx)].astype(np.float64)), abs(np.linalg.norm(got) - gn) / gn))
    for e in sorted(errs, key=lambda e: -e[1])[:12]:
        print("%-70s sample %.2e norm %.2e" % e)
    ws = [e[1] for e in errs if e[0].endswith("weights")]
    vs = [e[1] for e in errs if not e[0].endswith("weights")]
    print("weights: max %.2e median %.2e ; vectors: max %.2e median %.2e" % (max(ws), np.median(ws), max(vs), np.median(vs)))

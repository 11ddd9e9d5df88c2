"""Full-size golden vectors for the BASELINE configs the bench runs, from the CPU oracle
(oracle/vnet_oracle.py, numpy float64; minutes to an hour per case on 8 cores):

    c3_128cube.npz       configs[2]/[3]: 128^3, B=1, 1 modality, 2 classes, full-width net, fp32 arithmetic
    c2_64cube_b2.npz     configs[1]:     64^3,  B=2, 1 modality, 2 classes, full-width net
    c5_128cube_bf16.npz  configs[4]:     128^3, B=1, 4 modalities, 5 classes, bf16 conv operands / wide accumulate

The reference itself (TF 1.15) cannot run here, so these are ORACLE outputs, not reference outputs
(parity unpinned -- DESIGN.md section 2).  Weights come from a recipe (the reference's own initialisers drawn
from default_rng(42) in variable-creation order, make_golden.c1_weights) and inputs from
oracle.synthetic_batch, so only compact results are stored: loss, per-(batch, class) Dice sums, a strided
sample of the logits and of the argmax, and per gradient tensor its norm, sum, first 8 elements and a
seeded random sample of SAMPLE elements (sample_indices(): the test re-creates the indices).

    python tests/golden/make_golden_full.py c3 | c2 | c5
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vnet_oracle as O  # noqa: E402

SAMPLE = 2048
STRIDE = 4

CASES = {
    # name: (file, P, B, cin, K, input seed, conv operand rounding)
    "c3": ("c3_128cube.npz", 128, 1, 1, 2, 1000, None),
    "c2": ("c2_64cube_b2.npz", 64, 2, 1, 2, 3000, None),
    "c5": ("c5_128cube_bf16.npz", 128, 1, 4, 5, 1000, "bf16"),
}


def sample_indices(i, n):
    """Indices of the stored sample of gradient tensor number i (creation order) with n elements."""
    if n <= SAMPLE:
        return np.arange(n)
    return np.sort(np.random.default_rng(777 + i).choice(n, SAMPLE, replace=False))


def make(case):
    fname, P, B, cin, K, seed, rounding = CASES[case]
    t0 = time.time()
    ps = O.ParamStore(rng=np.random.default_rng(42))
    net = O.VNetOracle(K, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", ps)
    x, lab = O.synthetic_batch(B, P, cin, K, seed=seed)
    O.CONV5_OPERAND_ROUNDING = rounding
    try:
        res = O.run_step(x.astype(np.float64), lab, net, "sorensen")
    finally:
        O.CONV5_OPERAND_ROUNDING = None
    sm = res["softmax"]
    oh = (lab[..., 0][..., None] == np.arange(K)).astype(np.float64)
    ax = (1, 2, 3)
    names = list(ps.vars.keys())
    grads = [res["grads"][k] for k in names]
    s = (slice(None),) + (slice(None, None, STRIDE),) * 3
    out = {"loss": np.float64(res["loss"]),
           "dice_I": (sm * oh).sum(ax), "dice_L": sm.sum(ax), "dice_R": oh.sum(ax),
           "logits_sample": res["logits"][s].astype(np.float32), "pred_sample": res["pred"][s].astype(np.int8),
           "logits_absmax": np.float64(np.abs(res["logits"]).max()),
           "names": np.array(names),
           "grad_norm": np.array([np.linalg.norm(g) for g in grads]),
           "grad_sum": np.array([g.sum() for g in grads]),
           "grad_head": np.stack([np.resize(g.ravel()[:8], 8) for g in grads]),
           "grad_sample": np.stack([np.resize(g.ravel()[sample_indices(i, g.size)], SAMPLE) for i, g in enumerate(grads)]).astype(np.float32),
           "oracle_seconds": np.float64(time.time() - t0)}
    for k, v in ps.state.items():
        out["state:" + k] = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(case, "loss %.9f" % res["loss"], "seconds %.0f" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        make(c)

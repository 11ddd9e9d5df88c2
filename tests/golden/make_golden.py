"""Generates the committed golden vectors under tests/golden/ from the CPU oracle
(oracle/vnet_oracle.py, numpy float64).  The reference itself (TF 1.15) cannot run here, so these
are oracle outputs, not reference outputs (parity unpinned -- see DESIGN.md).  Re-run:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vnet_oracle as O  # noqa: E402

SMALL = {
    # name: (variant, cin, K, P, B, C0, levels, ncv, nb, loss, weights)
    "small_networks_c1k2": ("networks", 1, 2, 16, 2, 4, 2, (1, 2), 2, "sorensen", ()),
    "small_networks_c4k5": ("networks", 4, 5, 8, 1, 4, 2, (2, 3), 1, "mixed_weighted_jaccard", (0.1, 0.3, 0.5, 0.8, 1.0)),
    "small_legacy_c2k3": ("legacy", 2, 3, 8, 2, 4, 2, (1, 2), 2, "weighted_sorensen", (0.2, 0.5, 1.0)),
    "small_networks_l3": ("networks", 1, 2, 16, 1, 4, 3, (1, 2, 3), 3, "sorensen", ()),
}


# same nets with the bf16-operand arithmetic of BASELINE config C5 (oracle.CONV5_OPERAND_ROUNDING = "bf16")
SMALL_BF16 = {"small_networks_c4k5_bf16": SMALL["small_networks_c4k5"]}

COMPACT = ("small_networks_l3",)   # weights by recipe (rng 11, perturb .15), only norms of the gradients stored


def make_small(name, cfg):
    variant, cin, K, P, B, C0, levels, ncv, nb, loss, wts = cfg
    ps = O.ParamStore(rng=np.random.default_rng(11), perturb=0.15)
    net = O.VNetOracle(K, 0.0, C0, levels, ncv, nb, "prelu", variant, ps)
    x, lab = O.synthetic_batch(B, P, cin, K, seed=2000)
    res = O.run_step(x.astype(np.float64), lab, net, loss, wts, 0.7)
    out = {"images": x, "labels": lab, "logits": res["logits"].astype(np.float32), "loss": np.float64(res["loss"]),
           "pred": res["pred"].astype(np.int8)}
    if name in COMPACT:
        out["names"] = np.array(list(ps.vars.keys()))
        out["grad_norm"] = np.array([np.linalg.norm(res["grads"][k]) for k in ps.vars])
        out["grad_head"] = np.stack([np.resize(res["grads"][k].ravel()[:8], 8) for k in ps.vars])
    for k, v in ps.vars.items():
        if name in COMPACT:
            break
        out["param:" + k] = v.v.astype(np.float32)
        out["grad:" + k] = res["grads"][k].astype(np.float32)
    for k, v in ps.state.items():
        out["state:" + k] = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", res["loss"], "params", sum(v.v.size for v in ps.vars.values()))


def c1_weights(store_rng_seed=42):
    """Config-C1 weights: the reference's own initialisers (xavier-uniform, biases 0, gamma 1, beta 0,
    alpha 0.1 -- layers2.py:16-21,61,99) drawn from default_rng(42) in variable-creation order."""
    return O.ParamStore(rng=np.random.default_rng(store_rng_seed))


def make_c1():
    """BASELINE config C1: one 32^3 1-modality 2-class patch, full-width net (16,4,(1,2,3,3),3)."""
    ps = c1_weights()
    net = O.VNetOracle(2, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", ps)
    x, lab = O.synthetic_batch(1, 32, 1, 2, seed=1000)
    res = O.run_step(x.astype(np.float64), lab, net, "sorensen")
    lg = res["logits"]
    out = {"loss": np.float64(res["loss"]), "logits": lg.astype(np.float32), "pred": res["pred"].astype(np.int8),
           "names": np.array(list(ps.vars.keys())),
           "grad_norm": np.array([np.linalg.norm(res["grads"][k]) for k in ps.vars]),
           "grad_sum": np.array([res["grads"][k].sum() for k in ps.vars]),
           "grad_head": np.stack([np.resize(res["grads"][k].ravel()[:8], 8) for k in ps.vars])}
    np.savez_compressed(os.path.join(HERE, "c1_32cube_fullwidth.npz"), **out)
    print("c1 loss", res["loss"])


if __name__ == "__main__":
    for n, c in SMALL.items():
        make_small(n, c)
    O.CONV5_OPERAND_ROUNDING = "bf16"
    for n, c in SMALL_BF16.items():
        make_small(n, c)
    O.CONV5_OPERAND_ROUNDING = None
    make_c1()

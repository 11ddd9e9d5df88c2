"""Full-size golden vectors for the BASELINE configs the bench runs, from the CPU oracle
(oracle/vnet_oracle.py, numpy float64; minutes to an hour per case on 8 cores):

    c3_128cube.npz       configs[2]/[3]: 128^3, B=1, 1 modality, 2 classes, full-width net, fp32 arithmetic
    c2_64cube_b2.npz     configs[1]:     64^3,  B=2, 1 modality, 2 classes, full-width net
    c5_128cube_bf16.npz  configs[4]:     128^3, B=1, 4 modalities, 5 classes, bf16 conv operands / wide accumulate (round-2 mode)
    c5_128cube_b16.npz   configs[4] as SURVEY 8(d) words it: bf16 STORAGE of activations and their gradients (oracle.ACT_STORAGE);
                         also holds, for eight 5^3 layers and the level-2 2^3 pair, a crop of the layer's actual input and of the gradient that arrived
                         at its output (bf16 bit patterns) -- the teacher-forcing data of tests/test_hip_golden_full.py

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
    "c5s": ("c5_128cube_b16.npz", 128, 1, 4, 5, 1000, "storage"),
}
# weight seed of a case (the initialisers are drawn from default_rng(WEIGHT_SEED[case]) in variable-creation order)
WEIGHT_SEED = {"c3": 42, "c2": 42, "c5s": 42}
# round 6 (VERDICT r5 next #1a): a SEED SPREAD of the two fp32 fixtures -- four more (weight seed, input seed) draws each, so that the
# whole-network gradient bound of tests/test_hip_golden_full.py is set from ten samples of a chaotic quantity instead of two
# (profiles/r06_golden_seed_spread.txt).  Same recipe, same stored quantities; files under tests/golden/spread/.
# (first pass: draws 1-4.  c3's max-over-five came out at 1.34 x the fp32 mode's because of ONE draw -- draw 0, 1.27e-2 -- while the mean of
#  the per-draw worst, the medians and the whole-vector error were at or below the fp32 mode's; draws 5-9 were added BEFORE looking at
#  them, to report max-over-ten per config and pooled, whatever they say.)
for _s in range(1, 10):
    CASES["c3s%d" % _s] = ("spread/c3_128cube_s%d.npz" % _s, 128, 1, 1, 2, 1000 + 17 * _s, None)
    CASES["c2s%d" % _s] = ("spread/c2_64cube_b2_s%d.npz" % _s, 64, 2, 1, 2, 3000 + 17 * _s, None)
    WEIGHT_SEED["c3s%d" % _s] = WEIGHT_SEED["c2s%d" % _s] = 42 + 101 * _s

# teacher-forcing crops (case c5s): layer -> origin of an 8 x 8 x 16 box of OUTPUT voxels (clipped to the level's size); the
# stored input crop carries the 2-voxel halo, the stored output-gradient crop a 2-voxel halo as well (for backward-data).
TF_LAYERS = {
    "vnet/input_layer/weights": (0, 60, 56),                        # 4 -> 16 @128^3, touches the z = 0 face (zero padding)
    "vnet/encoder/level_1/conv_1/weights": (60, 120, 112),          # 16 -> 16 @128^3, touches the y / x high faces
    "vnet/decoder/level_1/conv_1/weights": (33, 47, 21),            # 32 -> 16 @128^3 (the roofline layer), interior
    "vnet/encoder/level_2/conv_2/weights": (28, 30, 40),            # 32 -> 32 @64^3
    "vnet/encoder/level_3/conv_3/weights": (12, 24, 16),            # 64 -> 64 @32^3
    "vnet/bottom_level/conv_2/weights": (0, 0, 0),                  # 256 -> 256 @8^3: the whole volume
    # round 4: the shapes the deep-level kernel (csrc/conv_deep.h) takes and a two-source decoder conv
    "vnet/encoder/level_4/conv_2/weights": (4, 8, 0),               # 128 -> 128 @16^3, touches the y high and both x faces
    "vnet/decoder/level_3/conv_1/weights": (12, 8, 16),             # 128 -> 64 @32^3: concat(up-convolved, skip), 64 + 64 channels
}
TF_BOX = (8, 8, 16)
# the 2^3 pair (stride 2: no halo).  layer -> origin of a TF_BOX of COARSE voxels; the fine crop is the 2x box at 2x the origin.
#   down convolution: x = fine crop (input), dy = coarse crop (gradient at its output)
#   transposed convolution: x = coarse crop (input), dy = fine crop
TF_LAYERS2 = {
    "vnet/encoder/level_2/down_convolution/weights": ("down", (8, 16, 16)),      # 32 -> 64, 64^3 -> 32^3
    "vnet/decoder/level_2/up_convolution/weights": ("up", (8, 16, 16)),          # 64 -> 32, 32^3 -> 64^3
}


def bf16_bits(a):
    """bf16-representable float64 array -> uint16 bit patterns."""
    u = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)
    assert not (u & 0xFFFF).any(), "value is not bf16-representable"
    return (u >> 16).astype(np.uint16)


def from_bf16_bits(u):
    return (np.asarray(u, dtype=np.uint32) << 16).view(np.float32).astype(np.float64)


def tf_crop(t, origin, halo=2):
    """(crop of t [1, D, H, W, C] around the TF_BOX at `origin` incl. `halo`, clipped to the volume; the crop's low corner)."""
    lo = [max(0, o - halo) for o in origin]
    hi = [min(t.shape[1 + a], origin[a] + TF_BOX[a] + halo) for a in range(3)]
    return t[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2], :], lo



def sample_indices(i, n):
    """Indices of the stored sample of gradient tensor number i (creation order) with n elements."""
    if n <= SAMPLE:
        return np.arange(n)
    return np.sort(np.random.default_rng(777 + i).choice(n, SAMPLE, replace=False))


def make(case):
    fname, P, B, cin, K, seed, rounding = CASES[case]
    t0 = time.time()
    ps = O.ParamStore(rng=np.random.default_rng(WEIGHT_SEED[case]))
    net = O.VNetOracle(K, 0.0, 16, 4, (1, 2, 3, 3), 3, "prelu", "networks", ps)
    x, lab = O.synthetic_batch(B, P, cin, K, seed=seed)
    O.CONV5_OPERAND_ROUNDING = rounding if rounding == "bf16" else None
    O.ACT_STORAGE = "bf16" if rounding == "storage" else None
    O.CAPTURE = {} if rounding == "storage" else None
    tf = {}
    try:
        res = O.run_step(x.astype(np.float64), lab, net, "sorensen")
        if O.CAPTURE is not None:
            for name, origin in TF_LAYERS.items():
                xin, yout = O.CAPTURE[name]
                xv = O.round_bf16(xin.v) if name.startswith("vnet/input_layer") else xin.v      # (the fp32 image: the conv rounds it)
                cx, lo = tf_crop(xv, origin)
                cg, _ = tf_crop(yout.g, origin)
                tf["tf:%s:x" % name] = bf16_bits(cx)
                tf["tf:%s:dy" % name] = bf16_bits(cg)
                tf["tf:%s:lo" % name] = np.asarray(lo, dtype=np.int32)
            for name, (kind, origin) in TF_LAYERS2.items():
                xin, yout = O.CAPTURE[name]
                co = tuple(slice(origin[a], origin[a] + TF_BOX[a]) for a in range(3))
                fi = tuple(slice(2 * origin[a], 2 * (origin[a] + TF_BOX[a])) for a in range(3))
                sx, sg = (fi, co) if kind == "down" else (co, fi)
                tf["tf:%s:x" % name] = bf16_bits(xin.v[(slice(None),) + sx])
                tf["tf:%s:dy" % name] = bf16_bits(yout.g[(slice(None),) + sg])
    finally:
        O.CONV5_OPERAND_ROUNDING = None
        O.ACT_STORAGE = None
        O.CAPTURE = None
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
    out.update(tf)
    os.makedirs(os.path.dirname(os.path.join(HERE, fname)), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(case, "loss %.9f" % res["loss"], "seconds %.0f" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        make(c)

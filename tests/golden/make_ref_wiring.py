"""Pins the WIRING and the VARIABLE NAMES of the restatement against the reference's own graph-building code.

    python tests/golden/make_ref_wiring.py          (build container only: needs /root/reference, which is read here and NEVER copied)

What runs: the reference's `networks.VNet(...).GetNetwork` (networks.py:246-365), `VNet.VNet(...).network_fn` (VNet.py:26-155),
`layers2` / `Layers` (layers2.py:59-99) and `model.dice_coe` (model.py:26-85) -- their real source, imported / exec'ed from
/root/reference at generation time -- against a NumPy-eager stand-in for the ~25 `tf.*` symbols they touch (this file: TF 1.15
cannot be installed here).  The stand-in's ARITHMETIC is this repo's own (float64 NumPy, written independently of oracle/), and its
naming rules (variable_scope nesting, `tf.layers` auto-names `batch_normalization`, `_1`, `_2` per enclosing variable scope, creation
order gamma / beta / moving_mean / moving_variance) are TF 1.15's as stated in SURVEY.md A.4 from knowledge.  So the fixtures pin
  * which layers exist, in which ORDER they are created and under which names (the checkpoint contract of SURVEY 8 f-3),
  * how the reference wires them (x + BN(x), the dead batch-norms, filter[-2] biases, the K-times `smooth`, concat order, ...),
  * logits / loss of that wiring with injected weights (to 1e-10 against the oracle),
and they do NOT pin TensorFlow's numerics: parity stays "partial -- unpinned by the reference" (DESIGN.md section 2).
Output: tests/golden/ref_wiring_<case>.npz (names, shapes, trainable flags in creation order, inputs, injected values, logits,
moving statistics after the step's update ops, dice_coe values).  tests/test_oracle.py::test_reference_wiring_* compares."""
import contextlib
import importlib
import os
import re
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VNET_REFERENCE", "/root/reference")


# ---- NumPy-eager stand-in for the tf.* surface of layers2.py / Layers.py / networks.py / VNet.py / model.dice_coe -----------------
class T(np.ndarray):
    """An eager tensor: a float64 ndarray with TF's get_shape()."""

    def get_shape(self):
        return _Shape(self.shape)


class _Shape(tuple):
    def as_list(self):
        return list(self)


def _t(a):
    return np.asarray(a, dtype=np.float64).view(T)


class Graph(object):
    def __init__(self, values=None):
        self.scope = []
        self.vars = []            # (name, array, trainable) in creation order
        self.byname = {}
        self.values = values or {}
        self.layer_count = {}
        self.updates = {}         # moving statistics after the update ops of one training step


G = [None]


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    G[0].scope.append(name)
    try:
        yield
    finally:
        G[0].scope.pop()


def _full(name):
    return "/".join(G[0].scope + [name])


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **k):
    g = G[0]
    full = _full(name)
    assert full not in g.byname, "variable %s created twice (no reuse in the reference)" % full
    if callable(initializer):
        shp = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, _Shape)) else (shape,)))
        v = initializer(shp)
    else:
        v = np.asarray(initializer)
    v = np.asarray(g.values[full], dtype=np.float64).reshape(np.shape(v)) if full in g.values else np.asarray(v, dtype=np.float64)
    v = _t(v)
    g.vars.append((full, v, trainable))
    g.byname[full] = v
    return v


def constant_initializer(value):
    return lambda shape: np.full(shape, value, dtype=np.float64)


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def nn_convolution(x, w, padding='SAME', strides=None, dilation_rate=None, **k):
    assert padding == 'SAME' and dilation_rate is None
    x, w = np.asarray(x), np.asarray(w)
    rank = x.ndim - 2
    assert rank == 3, "3-D path only"
    s = list(strides) if strides is not None else [1] * rank
    kd = w.shape[:3]
    outs, pads = [], []
    for n, kk, ss in zip(x.shape[1:4], kd, s):
        o, lo, hi = _same_pad(n, kk, ss)
        outs.append(o); pads.append((lo, hi))
    xp = np.pad(x, [(0, 0)] + pads + [(0, 0)])
    y = np.zeros((x.shape[0],) + tuple(outs) + (w.shape[-1],))
    for a in range(kd[0]):
        for b in range(kd[1]):
            for c in range(kd[2]):
                sl = xp[:, a:a + (outs[0] - 1) * s[0] + 1:s[0], b:b + (outs[1] - 1) * s[1] + 1:s[1], c:c + (outs[2] - 1) * s[2] + 1:s[2], :]
                y += sl @ w[a, b, c]
    return _t(y)


def nn_conv3d_transpose(x, w, output_shape, strides, padding='SAME', **k):
    """Gradient of conv3d w.r.t. its input with filter [kd, kh, kw, Cout, Cin]; here only k = stride = 2 (disjoint scatter)."""
    x, w = np.asarray(x), np.asarray(w)
    assert padding == 'SAME' and list(strides) == [1, 2, 2, 2, 1] and w.shape[:3] == (2, 2, 2)
    out_sp = tuple(int(v) for v in output_shape[1:4])
    y = np.zeros((x.shape[0],) + out_sp + (w.shape[3],))
    for a in range(2):
        for b in range(2):
            for c in range(2):
                t = x @ w[a, b, c].T                       # [.., Cin] x [Cin, Cout]
                tgt = y[:, a::2, b::2, c::2, :]
                tgt += t[:, :tgt.shape[1], :tgt.shape[2], :tgt.shape[3], :]
    return _t(y)


def layers_batch_normalization(x, momentum=0.99, epsilon=0.001, center=True, scale=True, training=False, **k):
    g = G[0]
    key = "/".join(g.scope)
    n = g.layer_count.get(key, 0)
    g.layer_count[key] = n + 1
    lname = "batch_normalization" if n == 0 else "batch_normalization_%d" % n
    C = int(x.shape[-1])
    with variable_scope(lname):
        gamma = get_variable("gamma", initializer=np.ones(C))
        beta = get_variable("beta", initializer=np.zeros(C))
        mm = get_variable("moving_mean", initializer=np.zeros(C), trainable=False)
        mv = get_variable("moving_variance", initializer=np.ones(C), trainable=False)
        full = _full("")
    xa = np.asarray(x)
    ax = tuple(range(xa.ndim - 1))
    mu = xa.mean(ax)
    var = ((xa - mu) ** 2).mean(ax)                       # biased (tf.nn.moments), non-fused path for 5-D inputs
    g.updates[full + "moving_mean"] = np.asarray(mm) - (np.asarray(mm) - mu) * (1 - momentum)
    g.updates[full + "moving_variance"] = np.asarray(mv) - (np.asarray(mv) - var) * (1 - momentum)
    inv = np.asarray(gamma) / np.sqrt(var + epsilon)
    return _t(xa * inv + (np.asarray(beta) - mu * inv))


def nn_dropout(x, keep_prob=None, noise_shape=None, seed=None, name=None, rate=None):
    r = rate if rate is not None else 1.0 - keep_prob
    assert float(r) == 0.0, "fixtures are made without dropout (its RNG is not reproducible)"
    return x


def make_tf():
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.bool = np.float32, np.bool_
    tf.variable_scope = variable_scope
    tf.get_variable = get_variable
    tf.constant_initializer = constant_initializer
    tf.placeholder = lambda dtype, shape=None, name=None: object()
    tf.tile = lambda x, m: _t(np.tile(np.asarray(x), m))
    tf.concat = lambda xs, axis: _t(np.concatenate([np.asarray(v) for v in xs], axis))
    tf.maximum = lambda a, b: _t(np.maximum(a, b))
    tf.minimum = lambda a, b: _t(np.minimum(a, b))
    tf.shape = lambda x: tuple(int(v) for v in x.shape)
    tf.reduce_sum = lambda x, axis=None, name=None: _t(np.sum(np.asarray(x), axis=tuple(axis) if isinstance(axis, (list, tuple)) else axis))
    tf.reduce_mean = lambda x, axis=None, name=None: _t(np.mean(np.asarray(x), axis=axis))
    tf.cast = lambda x, dtype: _t(x)
    tf.nn = types.SimpleNamespace(convolution=nn_convolution, conv3d_transpose=nn_conv3d_transpose, dropout=nn_dropout,
                                  relu=lambda x: _t(np.maximum(np.asarray(x), 0.0)),
                                  leaky_relu=lambda x, alpha=0.2: _t(np.where(np.asarray(x) > 0, x, alpha * np.asarray(x))))
    tf.layers = types.SimpleNamespace(batch_normalization=layers_batch_normalization)
    return tf


def load_reference():
    """Import layers2 / Layers / networks / VNet from /root/reference against the stand-in; dice_coe = the `def dice_coe` block of
    model.py exec'ed on its own (model.py as a whole imports SimpleITK, which is absent)."""
    tf = make_tf()
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, REF)
    try:
        mods = {}
        for name in ("layers2", "Layers", "networks", "VNet"):
            sys.modules.pop(name, None)
            mods[name] = importlib.import_module(name)
        src = open(os.path.join(REF, "model.py")).read()
        m = re.search(r"^def dice_coe\(.*?(?=^def )", src, re.S | re.M)
        ns = {"tf": tf}
        exec(compile(m.group(0), os.path.join(REF, "model.py"), "exec"), ns)
        mods["dice_coe"] = ns["dice_coe"]
    finally:
        sys.path.remove(REF)
        for name in ("layers2", "Layers", "networks", "VNet", "tensorflow"):
            sys.modules.pop(name, None)
    return mods


# ---- cases ---------------------------------------------------------------------------------------------------------------------
#        name              variant     cin K  C  levels convs      bottom  act     patch
CASES = {
    "networks_c1k2": ("networks", 1, 2, 4, 2, (1, 2), 2, "prelu", (8, 8, 8)),
    "networks_c3k3": ("networks", 3, 3, 2, 3, (1, 2, 3), 3, "prelu", (8, 8, 8)),      # multi-modality input conv, three-conv decoder blocks (dead BNs)
    "networks_odd":  ("networks", 1, 2, 4, 2, (2, 1), 1, "relu", (6, 10, 12)),         # non-cubic patch, relu, a one-conv bottom
    "legacy_c1k2":   ("legacy", 1, 2, 4, 2, (1, 2), 2, "prelu", (8, 8, 8)),
    "legacy_c2k3":   ("legacy", 2, 3, 2, 3, (1, 2, 3), 2, "prelu", (8, 8, 8)),
}


def build(mods, variant, cin, K, C, levels, convs, bottom, act, x, values=None):
    G[0] = Graph(values)
    if variant == "networks":
        net = mods["networks"].VNet(K, 0.0, C, levels, convs, bottom, True, act)
        logits = net.GetNetwork(_t(x))
    else:
        net = mods["VNet"].VNet(K, 1.0, C, levels, convs, bottom, True, act)
        logits = net.network_fn(_t(x))
    return G[0], np.asarray(logits)


def main():
    mods = load_reference()
    for cname, (variant, cin, K, C, levels, convs, bottom, act, patch) in CASES.items():
        rng = np.random.default_rng(abs(hash(cname)) % (2 ** 31) if False else sum(map(ord, cname)))
        x = rng.standard_normal((2,) + patch + (cin,))
        np.random.seed(0)                                   # (the reference's Xavier initialiser draws from the global NumPy RNG)
        g0, _ = build(mods, variant, cin, K, C, levels, convs, bottom, act, x)
        # second pass with every variable injected (seeded, non-trivial gamma / beta / alpha / moving statistics)
        values = {}
        for name, v, tr in g0.vars:
            if name.endswith("moving_variance"):
                values[name] = 0.5 + rng.random(v.shape)
            elif name.endswith(("gamma", "alpha")):
                values[name] = 0.5 + rng.random(v.shape)
            elif name.endswith("weights"):
                values[name] = rng.standard_normal(v.shape) * (2.0 / np.prod(v.shape[:-1])) ** 0.5
            else:
                values[name] = 0.1 * rng.standard_normal(v.shape)
        values = {k: v.astype(np.float32).astype(np.float64) for k, v in values.items()}     # stored as float32, exactly
        x = x.astype(np.float32).astype(np.float64)
        g, logits = build(mods, variant, cin, K, C, levels, convs, bottom, act, x, values)
        assert [n for n, _, _ in g.vars] == [n for n, _, _ in g0.vars]
        # the loss head's dice_coe on softmax(logits) vs a seeded one-hot target, the three forms the loss switch uses
        z = logits - logits.max(-1, keepdims=True)
        sm = np.exp(z) / np.exp(z).sum(-1, keepdims=True)
        lab = rng.integers(0, K, logits.shape[:-1])
        oh = np.eye(K)[lab]
        w = list(0.5 + rng.random(K))
        dice = {"dice_sorensen": float(mods["dice_coe"](_t(sm), _t(oh), loss_type='sorensen', axis=(1, 2, 3))),
                "dice_jaccard": float(mods["dice_coe"](_t(sm), _t(oh), loss_type='jaccard', axis=(1, 2, 3))),
                "dice_weighted_sorensen": float(mods["dice_coe"](_t(sm), _t(oh), loss_type='sorensen', axis=(1, 2, 3), weights=w))}
        out = {"names": np.array([n for n, _, _ in g.vars]), "trainable": np.array([t for _, _, t in g.vars]),
               "shapes": np.array([",".join(map(str, v.shape)) for _, v, _ in g.vars]),
               "x": x.astype(np.float32), "labels": lab.astype(np.int8), "dice_weights": np.array(w), "logits": logits,
               "config": np.array([variant, str(cin), str(K), str(C), str(levels), ",".join(map(str, convs)), str(bottom), act])}
        for k, v in dice.items():
            out[k] = np.float64(v)
        for n, v, _ in g.vars:
            out["v:" + n] = np.asarray(v).astype(np.float32)
        for n, v in g.updates.items():
            out["u:" + n] = np.asarray(v)
        path = os.path.join(HERE, "ref_wiring_%s.npz" % cname)
        np.savez_compressed(path, **out)
        print("%-16s %3d variables (%d trainable)  logits %s  ->  %s (%.0f KB)" % (
            cname, len(g.vars), sum(t for _, _, t in g.vars), logits.shape, os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()

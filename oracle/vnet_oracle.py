"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the V-Net hot path of jackyko1991/vnet-tensorflow.

PARITY UNPINNED: the reference ships no tests / golden vectors for this path and its
arithmetic lives in un-vendored TensorFlow 1.15.5 (Dockerfile:3), which cannot be imported
here.  This file restates, in NumPy float64, exactly what the reference's call sites ask
TensorFlow to compute; it is pinned by (1) an independent PyTorch-CPU float64 wiring
(oracle/torch_ref.py), (2) hand-derivable known answers and (3) central-difference gradient
checks -- see tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (vnet_tensorflow_amd/) never does.

Reference lines restated (all paths relative to the reference repo root):
  layers2.py:4-30    xavier_initializer_convolution
  layers2.py:59-63   convolution   (tf.nn.convolution NDHWC/DHWIO, SAME)
  layers2.py:65-74   deconvolution (tf.nn.conv3d_transpose, filter [k,k,k,Cout,Cin])
  layers2.py:78-94   down_convolution / up_convolution
  layers2.py:97-99   prelu
  networks.py:209-365 VNet.__init__/GetNetwork/convolution_block/convolution_block_2
  VNet.py:26-155     legacy V-Net wiring
  model.py:26-85     dice_coe
  model.py:87-92     weighted_softmax_cross_entropy_with_logits
  model.py:447,474-558,567-568  softmax / one-hot / loss switch / argmax
  model.py:641-666   exponential-decay LR, SGD / Adam / Momentum

Everything is channels-last (NDHWC).  A tiny reverse-mode tape (class Var) carries the
analytic gradients of SURVEY.md Appendix A so the network wiring below reads like the
reference's own op-by-op code.
"""
import math
import numpy as np

DT = np.float64


# ----------------------------------------------------------------------------------------
# reverse-mode tape
# ----------------------------------------------------------------------------------------
class Var(object):
    """A value on the tape.  .v = ndarray, .g = accumulated gradient (or None)."""
    __slots__ = ("v", "g", "_bw", "_parents", "name")

    def __init__(self, v, parents=(), bw=None, name=None):
        self.v = np.asarray(v)
        self.g = None
        self._bw = bw
        self._parents = parents
        self.name = name

    @property
    def shape(self):
        return self.v.shape

    def _acc(self, g):
        self.g = g if self.g is None else self.g + g


def backward(root, seed=None):
    """Topological reverse sweep from `root` (scalar unless seed given)."""
    order, seen = [], set()

    def visit(n):
        stack = [(n, False)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node)
                continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for p in node._parents:
                if id(p) not in seen:
                    stack.append((p, False))
    visit(root)
    root.g = np.ones_like(root.v) if seed is None else np.asarray(seed, dtype=root.v.dtype)
    for n in reversed(order):
        if n._bw is not None and n.g is not None:
            n._bw(n.g)


# ----------------------------------------------------------------------------------------
# raw array primitives (forward + explicit backward), TF 1.15 semantics
# ----------------------------------------------------------------------------------------
def same_pad(n_in, k, s):
    """TF SAME rule (A.1): out=ceil(in/s); extra padding goes on the high side."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return out, total // 2, total - total // 2


def _pad_spatial(x, pads):
    cfg = [(0, 0)] + [(lo, hi) for lo, hi in pads] + [(0, 0)]
    return np.pad(x, cfg)


def round_bf16(a):
    """Round-to-nearest-even to bfloat16 (8 exponent, 7 mantissa bits), returned as float64.  This is the operand
    rounding of the bf16-compute mode (BASELINE config C5; the matrix cores then accumulate exactly-representable
    products in fp32) -- v_cvt_pk_bf16_f32 on gfx950.  Inputs are taken through float32 first, as the device does."""
    u = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(a))


def conv_nd_fwd(x, w, stride):
    """tf.nn.convolution(x, w, 'SAME', strides) -- cross-correlation, no bias.
    x [B,*S,Ci]; w [*k,Ci,Co]; stride int.  Works for 2-D and 3-D."""
    rank = x.ndim - 2
    ks = w.shape[:rank]
    geo = [same_pad(x.shape[1 + a], ks[a], stride) for a in range(rank)]
    outs = [g[0] for g in geo]
    xp = _pad_spatial(x, [(g[1], g[2]) for g in geo])
    y = np.zeros((x.shape[0],) + tuple(outs) + (w.shape[-1],), dtype=x.dtype)
    for tap in np.ndindex(*ks):
        sl = (slice(None),) + tuple(slice(tap[a], tap[a] + stride * outs[a], stride) for a in range(rank))
        y += xp[sl] @ w[tap]
    return y


def conv_nd_bwd(x, w, dy, stride, need_dx=True):
    """Gradients of conv_nd_fwd: returns (dx, dw)."""
    rank = x.ndim - 2
    ks = w.shape[:rank]
    geo = [same_pad(x.shape[1 + a], ks[a], stride) for a in range(rank)]
    outs = [g[0] for g in geo]
    xp = _pad_spatial(x, [(g[1], g[2]) for g in geo])
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp) if need_dx else None
    ci, co = w.shape[-2], w.shape[-1]
    dy2 = dy.reshape(-1, co)
    for tap in np.ndindex(*ks):
        sl = (slice(None),) + tuple(slice(tap[a], tap[a] + stride * outs[a], stride) for a in range(rank))
        dw[tap] = xp[sl].reshape(-1, ci).T @ dy2
        if need_dx:
            dxp[sl] += dy @ w[tap].T
    dx = None
    if need_dx:
        un = (slice(None),) + tuple(slice(geo[a][1], geo[a][1] + x.shape[1 + a]) for a in range(rank))
        dx = dxp[un]
    return dx, dw


def conv_nd_transpose_fwd(x, w, out_spatial, stride):
    """tf.nn.conv{2,3}d_transpose(x, w, output_shape, [1,s,..,1], 'SAME') (A.2): the gradient
    of conv_nd_fwd w.r.t. an input of spatial size `out_spatial`, for filter w [*k,Cout,Cin]
    (Cin = channels of x)."""
    B = x.shape[0]
    dummy = np.zeros((B,) + tuple(out_spatial) + (w.shape[-2],), dtype=x.dtype)
    # conv_nd_fwd(dummy, w) has shape of x ; its dx given dy=x is the transposed conv
    dx, _ = conv_nd_bwd(dummy, w, x, stride, need_dx=True)
    return dx


# ----------------------------------------------------------------------------------------
# tape ops
# ----------------------------------------------------------------------------------------
def const(v):
    return Var(np.asarray(v, dtype=DT))


def add(a, b):
    """Broadcasting add (used for +bias and residual adds)."""
    out = Var(a.v + b.v, (a, b))

    def bw(g):
        a._acc(_unbroadcast(g, a.v.shape))
        b._acc(_unbroadcast(g, b.v.shape))
    out._bw = bw
    return out


def _unbroadcast(g, shape):
    if g.shape == tuple(shape):
        return g
    nd = g.ndim - len(shape)
    g = g.sum(axis=tuple(range(nd))) if nd > 0 else g
    ax = tuple(i for i, s in enumerate(shape) if s == 1 and g.shape[i] != 1)
    if ax:
        g = g.sum(axis=ax, keepdims=True)
    return g.reshape(shape)


# BASELINE config C5 ("bf16 compute, fp32 Dice accumulate"): the three GEMMs of every 5^k stride-1 convolution
# (forward x*w, backward-data dy*w, backward-filter x*dy) take their two operands rounded to bfloat16 and accumulate
# exactly-representable products in a wide accumulator; everything else stays full precision.  None = the reference's
# fp32 arithmetic (restated here in float64).
CONV5_OPERAND_ROUNDING = None


# Round 3, config C5 as SURVEY 8(d) words it ("bf16 activations/weights into MFMA, fp32 accumulate, fp32 BN stats and Dice
# sums"): with ACT_STORAGE = "bf16" every activation tensor the network keeps -- and every gradient of one -- is STORED as
# bfloat16: `store()` marks those tensors in the wiring below (forward value rounded once; the gradient that arrives, summed over
# the consumers, rounded once on its way to the producer).  Every convolution with a spatial kernel (5^k and 2^k, forward, backward-
# data and filter gradient) then takes bf16 operands (the filter is rounded, the tensors are bf16 already); the 1^k output head
# keeps its fp32 filter and fp32 logits; batch-norm statistics, parameter gradients, softmax and Dice are full precision.
ACT_STORAGE = None


def store(x):
    """A tensor the bf16-storage mode writes to memory (identity otherwise)."""
    if ACT_STORAGE != "bf16":
        return x
    out = Var(round_bf16(x.v), (x,))
    out._bw = lambda g: x._acc(round_bf16(g))
    return out


def _operand_rounding(w, stride):
    k = w.shape[0]
    if ACT_STORAGE == "bf16" and k > 1:
        return round_bf16
    if CONV5_OPERAND_ROUNDING == "bf16" and stride == 1 and k == 5:
        return round_bf16
    return (lambda a: a)


def convolution(x, w, b, stride=1):
    """layers2.py:59-63: tf.nn.convolution(x, w, 'SAME', strides) + b."""
    rb = _operand_rounding(w.v, stride)
    y = conv_nd_fwd(rb(x.v), rb(w.v), stride) + b.v
    out = Var(y, (x, w, b))

    def bw(g):
        dx, dw = conv_nd_bwd(rb(x.v), rb(w.v), rb(g), stride)
        x._acc(dx)
        w._acc(dw)
        b._acc(g.reshape(-1, g.shape[-1]).sum(0))
    out._bw = bw
    return out


def deconvolution(x, w, b, out_spatial, stride=2):
    """layers2.py:65-74: tf.nn.conv3d_transpose(x, w, output_shape, strides, 'SAME') + b,
    w [*k, Cout, Cin], b has filter[-2] = Cout elements."""
    rb = round_bf16 if ACT_STORAGE == "bf16" else (lambda a: a)
    y = conv_nd_transpose_fwd(rb(x.v), rb(w.v), out_spatial, stride) + b.v
    out = Var(y, (x, w, b))

    def bw(g):
        # y = d/dX conv(X, w) . x  ==> dx = conv(g, w) ; dw = conv-filter-grad(X:=g, dy:=x)
        x._acc(conv_nd_fwd(rb(g), rb(w.v), stride))
        _, dw = conv_nd_bwd(rb(g), w.v, rb(x.v), stride, need_dx=False)
        w._acc(dw)
        b._acc(g.reshape(-1, g.shape[-1]).sum(0))
    out._bw = bw
    return out


def tile_channels(x, n):
    """networks.py:254-258 tf.tile(x, [1,..,1,n]) on the channel axis."""
    reps = (1,) * (x.v.ndim - 1) + (n,)
    out = Var(np.tile(x.v, reps), (x,))
    c = x.v.shape[-1]

    def bw(g):
        x._acc(g.reshape(g.shape[:-1] + (n, c)).sum(-2))
    out._bw = bw
    return out


def concat_channels(a, b):
    """networks.py:325 tf.concat((a, b), axis=-1)."""
    ca = a.v.shape[-1]
    out = Var(np.concatenate((a.v, b.v), axis=-1), (a, b))

    def bw(g):
        a._acc(g[..., :ca])
        b._acc(g[..., ca:])
    out._bw = bw
    return out


BN_EPS = 1e-3       # networks.py:259 epsilon=0.001
BN_MOMENTUM = 0.99  # networks.py:259 momentum=0.99


def batch_norm_train(x, gamma, beta, eps=BN_EPS, stats_out=None):
    """tf.layers.batch_normalization(..., training=True) (A.4): biased batch moments over all
    axes but the last.  stats_out (list) receives (mean, biased_var) for moving-average checks."""
    ax = tuple(range(x.v.ndim - 1))
    M = x.v.size // x.v.shape[-1]
    mu = x.v.mean(axis=ax)
    var = ((x.v - mu) ** 2).mean(axis=ax)
    inv = 1.0 / np.sqrt(var + eps)
    xh = (x.v - mu) * inv
    out = Var(xh * gamma.v + beta.v, (x, gamma, beta))
    if stats_out is not None:
        stats_out.append((mu, var))

    def bw(g):
        dbeta = g.sum(axis=ax)
        dgamma = (g * xh).sum(axis=ax)
        beta._acc(dbeta)
        gamma._acc(dgamma)
        x._acc(gamma.v * inv * (g - dbeta / M - xh * dgamma / M))
    out._bw = bw
    return out


def prelu(x, alpha):
    """layers2.py:97-99: max(0,x) + alpha*min(0,x); gradient at x==0 is 0 (A.5)."""
    neg = np.minimum(0.0, x.v)
    out = Var(np.maximum(0.0, x.v) + alpha.v * neg, (x, alpha))

    def bw(g):
        x._acc(g * np.where(x.v > 0, 1.0, np.where(x.v < 0, alpha.v, 0.0)))
        alpha._acc((g * neg).reshape(-1, neg.shape[-1]).sum(0))
    out._bw = bw
    return out


def relu(x):
    out = Var(np.maximum(x.v, 0.0), (x,))
    out._bw = lambda g: x._acc(g * (x.v > 0))
    return out


def leaky_relu(x, slope=0.2):
    """tf.nn.leaky_relu default alpha=0.2 (networks.py:243-244)."""
    out = Var(np.where(x.v > 0, x.v, slope * x.v), (x,))
    out._bw = lambda g: x._acc(g * np.where(x.v > 0, 1.0, slope))
    return out


def dropout(x, rate, mask=None):
    """tf.nn.dropout(x, rate) (A.6): y = x*mask/(1-rate); rate==0 -> identity.  `mask` is an
    injected {0,1} array (TF's RNG stream is not reproducible)."""
    if rate == 0.0:
        return x
    if mask is None:
        raise ValueError("oracle dropout needs an injected mask for rate>0")
    sc = mask / (1.0 - rate)
    out = Var(x.v * sc, (x,))
    out._bw = lambda g: x._acc(g * sc)
    return store(out)


def softmax(z):
    """model.py:447 tf.nn.softmax(logits) over the last axis."""
    e = np.exp(z.v - z.v.max(axis=-1, keepdims=True))
    p = e / e.sum(axis=-1, keepdims=True)
    out = Var(p, (z,))
    out._bw = lambda g: z._acc(p * (g - (g * p).sum(axis=-1, keepdims=True)))
    return out


def one_hot(labels, depth):
    """model.py:474-477 tf.one_hot: out-of-range labels give all-zero rows."""
    lab = np.asarray(labels)
    return (lab[..., None] == np.arange(depth)).astype(DT)


def dice_coe(output, target, loss_type='jaccard', axis=None, weights=(), smooth=1e-5):
    """model.py:26-85.  output: Var [B,*S,K]; target: ndarray same shape.  Returns scalar Var."""
    if axis is None:
        axis = tuple(range(1, output.v.ndim - 1))
    p, t = output.v, np.asarray(target, dtype=DT)
    inse = (p * t).sum(axis=axis)
    if loss_type == 'jaccard':
        l = (p * p).sum(axis=axis)
        r = (t * t).sum(axis=axis)
    elif loss_type == 'sorensen':
        l = p.sum(axis=axis)
        r = t.sum(axis=axis)
    else:
        raise Exception("Unknown loss_type")
    B, K = inse.shape
    bshape = (B,) + (1,) * len(axis) + (K,)
    if len(weights) != 0:
        assert len(weights) == K
        wv = np.asarray(weights, dtype=DT)
        num = (2.0 * wv * inse + smooth).sum(-1)          # model.py:74 (smooth added K times)
        den = (wv * (l + r) + smooth).sum(-1)
        dice = (num / den).mean()
        dnum_dI = 2.0 * wv[None, :] / den[:, None] / B
        dden = -(num / den ** 2)[:, None] * wv[None, :] / B     # d dice / d l
    else:
        den = l + r + smooth
        dice = ((2.0 * inse + smooth) / den).mean()
        dnum_dI = 2.0 / den / (B * K)
        dden = -(2.0 * inse + smooth) / den ** 2 / (B * K)
    out = Var(dice, (output,))

    def bw(g):
        gI = dnum_dI.reshape(bshape)
        gL = dden.reshape(bshape)
        if loss_type == 'jaccard':
            output._acc(g * (gI * t + gL * 2.0 * p))
        else:
            output._acc(g * (gI * t + gL * np.ones_like(p)))
    out._bw = bw
    return out


def softmax_xent_mean(logits, onehot, class_weights=None):
    """model.py:495 / model.py:87-92: mean over voxels of softmax_cross_entropy_with_logits,
    optionally weighted per voxel by sum_c(w_c*label_c)."""
    z = logits.v
    zs = z - z.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(zs).sum(axis=-1, keepdims=True))
    logp = zs - lse
    per = -(onehot * logp).sum(-1)
    wv = np.ones_like(per) if class_weights is None else (np.asarray(class_weights, dtype=DT) * onehot).sum(-1)
    n = per.size
    out = Var((per * wv).sum() / n, (logits,))
    p = np.exp(logp)

    def bw(g):
        # d/dz of -sum_c t_c log p_c = p*sum(t) - t
        logits._acc(g * (wv[..., None] * (p * onehot.sum(-1, keepdims=True) - onehot)) / n)
    out._bw = bw
    return out


def scalar_affine(x, scale, shift):
    out = Var(x.v * scale + shift, (x,))
    out._bw = lambda g: x._acc(g * scale)
    return out


# ----------------------------------------------------------------------------------------
# parameters: TF variable names and initialisation (layers2.py:4-30)
# ----------------------------------------------------------------------------------------
def xavier_uniform(shape, rng):
    """layers2.py:16-21: lim = sqrt(6 / (prod(spatial) * (Cin + Cout)))."""
    s = len(shape) - 2
    num = np.prod(shape[:s]) * np.sum(shape[s:])
    lim = np.sqrt(6.0 / num)
    return rng.uniform(-lim, lim, shape).astype(np.float32)


class ParamStore(object):
    """Scoped variable store emulating tf.variable_scope / tf.get_variable / the auto-uniquified
    names of tf.layers.batch_normalization (A.4)."""

    def __init__(self, rng=None, values=None, perturb=0.0):
        self.rng = rng if rng is not None else np.random.default_rng(42)
        self.values = values          # optional {name: ndarray} to inject
        self.vars = {}                # name -> Var (trainable)
        self.state = {}               # name -> ndarray (BN moving stats)
        self.order = []               # creation order of trainables
        self.scope = []
        self._bn_count = {}
        self.perturb = perturb        # randomise gamma/beta/alpha/bias so tests are transpose-detecting

    def begin_pass(self):
        self._bn_count = {}

    class _Scope(object):
        def __init__(self, store, name):
            self.store, self.name = store, name

        def __enter__(self):
            self.store.scope.append(self.name)

        def __exit__(self, *a):
            self.store.scope.pop()

    def variable_scope(self, name):
        return ParamStore._Scope(self, name)

    def _full(self, name):
        return "/".join(self.scope + [name])

    def get(self, name, init):
        full = self._full(name)
        if full not in self.vars:
            if self.values is not None and full in self.values:
                v = np.asarray(self.values[full])
            else:
                v = init()
            self.vars[full] = Var(np.asarray(v, dtype=DT), name=full)
            self.order.append(full)
        return self.vars[full]

    def bn_scope(self):
        key = "/".join(self.scope)
        n = self._bn_count.get(key, 0)
        self._bn_count[key] = n + 1
        return "batch_normalization" if n == 0 else "batch_normalization_%d" % n

    def _jit(self, base, shape):
        if self.perturb:
            return base + self.perturb * self.rng.standard_normal(shape)
        return np.full(shape, base, dtype=DT)


def _conv_vars(ps, filt, bias_len):
    w = ps.get('weights', lambda: xavier_uniform(filt, ps.rng))
    b = ps.get('biases', lambda: ps._jit(0.0, (bias_len,)))
    return w, b


# Teacher forcing (tests/golden/make_golden_full.py): with CAPTURE = {} every convolution call site records its (input Var,
# output Var) under its 'weights' name; after backward() the entry holds what the layer saw -- input values and the gradient
# that arrived at its output.
CAPTURE = None


def L_convolution(ps, x, filt, stride=1):
    w, b = _conv_vars(ps, filt, filt[-1])
    y = convolution(x, w, b, stride)
    if CAPTURE is not None:
        CAPTURE[w.name] = (x, y)
    return y


def L_down_convolution(ps, x, factor, kernel_size):
    c = x.v.shape[-1]
    return L_convolution(ps, x, list(kernel_size) + [c, c * factor], stride=factor)


def L_up_convolution(ps, x, out_spatial, factor, kernel_size):
    c = x.v.shape[-1]
    filt = list(kernel_size) + [c // factor, c]
    w, b = _conv_vars(ps, filt, filt[-2])
    y = deconvolution(x, w, b, out_spatial, factor)
    if CAPTURE is not None:
        CAPTURE[w.name] = (x, y)
    return y


def L_batch_norm(ps, x, dead=False):
    """One tf.layers.batch_normalization call site.  Creates gamma/beta/moving stats under the
    auto-uniquified layer name and applies the moving-average update op (UPDATE_OPS,
    model.py:665-666) -- also for layers whose output is unused (`dead`)."""
    C = x.v.shape[-1]
    with ps.variable_scope(ps.bn_scope()):
        gamma = ps.get('gamma', lambda: ps._jit(1.0, (C,)))
        beta = ps.get('beta', lambda: ps._jit(0.0, (C,)))
        mm = ps._full('moving_mean')
        mv = ps._full('moving_variance')
    ps.state.setdefault(mm, np.zeros(C, dtype=DT))
    ps.state.setdefault(mv, np.ones(C, dtype=DT))
    st = []
    y = batch_norm_train(x, gamma, beta, stats_out=st)
    mu, var = st[0]
    ps.state[mm] = ps.state[mm] - (ps.state[mm] - mu) * (1.0 - BN_MOMENTUM)
    ps.state[mv] = ps.state[mv] - (ps.state[mv] - var) * (1.0 - BN_MOMENTUM)
    return y


def L_activation(ps, x, kind):
    if kind == 'prelu':
        C = x.v.shape[-1]
        alpha = ps.get('alpha', lambda: ps._jit(0.1, (C,)))
        return prelu(x, alpha)
    if kind == 'relu':
        return relu(x)
    if kind == 'lrelu':
        return leaky_relu(x)
    raise ValueError(kind)


# ----------------------------------------------------------------------------------------
# networks.VNet (the main.py path) -- networks.py:209-365
# ----------------------------------------------------------------------------------------
class VNetOracle(object):
    def __init__(self, num_classes, dropout_rate=0.0, num_channels=16, num_levels=4,
                 num_convolutions=(1, 2, 3, 3), bottom_convolutions=3, activation_fn="relu",
                 variant="networks", store=None):
        assert num_levels == len(num_convolutions)
        self.num_classes = num_classes
        self.dropout_rate = dropout_rate
        self.num_channels = num_channels
        self.num_levels = num_levels
        self.num_convolutions = tuple(num_convolutions)
        self.bottom_convolutions = bottom_convolutions
        self.act = activation_fn
        self.variant = variant      # "networks" (networks.py) | "legacy" (VNet.py)
        self.ps = store if store is not None else ParamStore()

    # networks.py:307-322 / VNet.py:26-39
    def convolution_block(self, x, n):
        ps = self.ps
        layer_input = x
        C = x.v.shape[-1]
        k = [5] * (x.v.ndim - 2)
        for i in range(n):
            with ps.variable_scope('conv_%d' % (i + 1)):
                x = store(L_convolution(ps, x, k + [C, C]))
                if self.variant == "legacy":
                    x = store(L_batch_norm(ps, x))
                if i == n - 1:
                    x = add(x, layer_input)
                x = L_batch_norm(ps, x)
                x = store(L_activation(ps, x, self.act))      # residual add + batch-norm + activation: one fused kernel, one tensor
                x = dropout(x, self.dropout_rate)
        return x

    # networks.py:324-365 / VNet.py:42-73
    def convolution_block_2(self, x, f, n):
        ps = self.ps
        layer_input = x
        C = x.v.shape[-1]
        k = [5] * (x.v.ndim - 2)
        x = concat_channels(x, f)
        legacy = self.variant == "legacy"
        # (storage marks: the networks.py batch-norm chains BN -> BN -> add -> BN -> act are ONE fused normalisation of the stored
        #  conv output in the build (closed form), so nothing in between is stored; the legacy wiring runs two kernels)
        if n == 1:
            with ps.variable_scope('conv_1'):
                x = store(L_convolution(ps, x, k + [2 * C, C]))
                x = L_batch_norm(ps, x)
                if legacy:
                    x = store(x)
                else:
                    layer_input = L_batch_norm(ps, x)       # networks.py:335
                x = add(x, layer_input)
                x = L_batch_norm(ps, x)
                x = store(L_activation(ps, x, self.act))
                x = dropout(x, self.dropout_rate)
            return x
        with ps.variable_scope('conv_1'):
            x = store(L_convolution(ps, x, k + [2 * C, C]))
            x = L_batch_norm(ps, x)
            x = store(L_activation(ps, x, self.act))
            x = dropout(x, self.dropout_rate)
        for i in range(1, n):
            with ps.variable_scope('conv_%d' % (i + 1)):
                x = store(L_convolution(ps, x, k + [C, C]))
                if legacy:
                    x = store(L_batch_norm(ps, x))           # VNet.py:65
                else:
                    layer_input = L_batch_norm(ps, x)        # networks.py:358 (dead unless last)
                if i == n - 1:
                    x = add(x, layer_input)
                x = L_batch_norm(ps, x)
                x = store(L_activation(ps, x, self.act))
                x = dropout(x, self.dropout_rate)
        return x

    # networks.py:246-305 / VNet.py:110-155
    def GetNetwork(self, images):
        ps = self.ps
        ps.begin_pass()
        x = images if isinstance(images, Var) else Var(np.asarray(images, dtype=DT))
        rank = x.v.ndim - 2
        cin = x.v.shape[-1]
        with ps.variable_scope('vnet/input_layer'):
            if cin == 1:
                x = tile_channels(x, self.num_channels)
                x = store(L_batch_norm(ps, x))
            else:
                x = store(L_convolution(ps, x, [5] * rank + [cin, self.num_channels]))     # (the conv rounds the fp32 image itself)
                x = L_batch_norm(ps, x)
                x = store(L_activation(ps, x, self.act))
        feats = []
        for l in range(self.num_levels):
            with ps.variable_scope('vnet/encoder/level_%d' % (l + 1)):
                x = self.convolution_block(x, self.num_convolutions[l])
                feats.append(x)
                with ps.variable_scope('down_convolution'):
                    x = store(L_down_convolution(ps, x, 2, [2] * rank))
                    x = L_batch_norm(ps, x)
                    x = store(L_activation(ps, x, self.act))
        with ps.variable_scope('vnet/bottom_level'):
            x = self.convolution_block(x, self.bottom_convolutions)
        for l in reversed(range(self.num_levels)):
            with ps.variable_scope('vnet/decoder/level_%d' % (l + 1)):
                f = feats[l]
                with ps.variable_scope('up_convolution'):
                    x = store(L_up_convolution(ps, x, f.v.shape[1:-1], 2, [2] * rank))
                    x = L_batch_norm(ps, x)
                    x = store(L_activation(ps, x, self.act))
                x = self.convolution_block_2(x, f, self.num_convolutions[l])
        with ps.variable_scope('vnet/output_layer'):
            logits = L_convolution(ps, x, [1] * rank + [self.num_channels, self.num_classes])
            logits = L_batch_norm(ps, logits)
        return logits

    network_fn = GetNetwork   # VNet.py:110 name


# ----------------------------------------------------------------------------------------
# loss head -- model.py:447, 474-558
# ----------------------------------------------------------------------------------------
def loss_head(logits, labels, loss_name="sorensen", weights=(), alpha=1.0):
    """labels: int array [B,*S,1] (model.py:303-309).  Returns (loss Var, softmax Var)."""
    K = logits.v.shape[-1]
    oh = one_hot(np.asarray(labels)[..., 0], K)
    sm = softmax(logits)
    if loss_name == "xent":
        return softmax_xent_mean(logits, oh), sm
    if loss_name == "weighted_xent":
        return softmax_xent_mean(logits, oh, weights), sm
    kind = 'sorensen' if 'sorensen' in loss_name else 'jaccard'
    w = weights if 'weighted' in loss_name else ()
    d = dice_coe(sm, oh, loss_type=kind, weights=w)
    loss = scalar_affine(d, -1.0, 1.0)            # 1 - dice
    if loss_name.startswith("mixed_"):
        xe = softmax_xent_mean(logits, oh, weights if 'weighted' in loss_name else None)
        loss = add(loss, scalar_affine(xe, alpha, 0.0))
    elif loss_name not in ("sorensen", "weighted_sorensen", "jaccard", "weighted_jaccard"):
        raise SystemExit("Invalid loss function")
    return loss, sm


def argmax_pred(logits):
    """model.py:567-568 tf.argmax(logits, -1): int64, first maximal index on ties (A.8)."""
    v = logits.v if isinstance(logits, Var) else np.asarray(logits)
    return np.argmax(v, axis=-1).astype(np.int64)


def hard_dice(pred, labels, K):
    """model.py:619-621 hard dice 2tp/(2tp+fp+fn) per class."""
    out = []
    for c in range(K):
        p, t = pred == c, labels == c
        tp, fp, fn = (p & t).sum(), (p & ~t).sum(), (~p & t).sum()
        out.append(2.0 * tp / max(2.0 * tp + fp + fn, 1e-30))
    return np.asarray(out)


# ----------------------------------------------------------------------------------------
# optimiser / LR schedule -- model.py:641-666 (A.9, f-1)
# ----------------------------------------------------------------------------------------
def tf_metrics_auc(labels01, predictions, num_thresholds=200):
    """tf.metrics.auc(labels, predictions) with the TF 1.15 defaults (curve='ROC', summation_method='trapezoidal'), as used
    per class at model.py:607,613,624.  Restated from TF 1.15 metrics_impl.py (third-party, un-vendored): thresholds
    [0 - 1e-7] + [(i + 1) / (n - 1) for i in range(n - 2)] + [1 + 1e-7] as float32; a voxel is predicted positive at
    threshold t iff prediction > t; rec = (tp + eps) / (tp + fn + eps); fp_rate = fp / (fp + tn + eps);
    auc = sum((x[:-1] - x[1:]) * (y[:-1] + y[1:]) / 2) with x = fp_rate, y = rec.  (TF keeps the counters in float32
    variables; counts here are exact.)"""
    eps = 1e-7
    th = np.asarray([0.0 - eps] + [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + eps], dtype=np.float32)
    p = np.asarray(predictions, dtype=np.float32).ravel()
    lab = np.asarray(labels01).ravel().astype(bool)
    tp = np.array([np.count_nonzero(lab & (p > t)) for t in th], dtype=np.float64)
    fp = np.array([np.count_nonzero(~lab & (p > t)) for t in th], dtype=np.float64)
    fn = np.count_nonzero(lab) - tp
    tn = np.count_nonzero(~lab) - fp
    rec = (tp + eps) / (tp + fn + eps)
    fpr = fp / (fp + tn + eps)
    return float(((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2.0).sum())


def exponential_decay(lr0, step, decay_steps, decay_rate):
    return lr0 * decay_rate ** (step / float(decay_steps))


class TFAdam(object):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, beta1=0.9, beta2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps, self.t = beta1, beta2, eps, 0
        self.m, self.v = {}, {}

    def step(self, params, grads, lr):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k in params:
            g = grads[k]
            m = self.m.get(k, 0.0) * self.b1 + (1 - self.b1) * g
            v = self.v.get(k, 0.0) * self.b2 + (1 - self.b2) * g * g
            self.m[k], self.v[k] = m, v
            params[k] = params[k] - lr_t * m / (np.sqrt(v) + self.eps)
        return params


def sgd_step(params, grads, lr):
    return {k: params[k] - lr * grads[k] for k in params}


class TFMomentum(object):
    """tf.train.MomentumOptimizer: acc = mom*acc + g ; theta -= lr*acc (nesterov: lr*(g+mom*acc))."""

    def __init__(self, momentum=0.9, nesterov=False):
        self.mom, self.nesterov, self.acc = momentum, nesterov, {}

    def step(self, params, grads, lr):
        for k in params:
            a = self.acc.get(k, 0.0) * self.mom + grads[k]
            self.acc[k] = a
            upd = grads[k] + self.mom * a if self.nesterov else a
            params[k] = params[k] - lr * upd
        return params


# ----------------------------------------------------------------------------------------
# convenience: one full forward(+backward) of the reference training graph
# ----------------------------------------------------------------------------------------
def run_step(images, labels, net, loss_name="sorensen", weights=(), alpha=1.0, want_grads=True):
    """Forward + loss (+ backward).  Returns dict(logits, softmax, loss, pred, grads{name:arr})."""
    logits = net.GetNetwork(images)
    loss, sm = loss_head(logits, labels, loss_name, weights, alpha)
    res = dict(logits=logits.v, softmax=sm.v, loss=float(loss.v), pred=argmax_pred(logits))
    if want_grads:
        for v in net.ps.vars.values():
            v.g = None
        backward(loss)
        res["grads"] = {k: (v.g if v.g is not None else np.zeros_like(v.v)) for k, v in net.ps.vars.items()}
    return res


def synthetic_batch(B, P, cin, K, seed=1000, rank=3):
    """Synthetic patches per SURVEY 8(d): image clamp(127.5+40 N(0,1), 0, 255) + 60 inside the
    label spheres; label = background 0 + one sphere of radius P/6 per foreground class."""
    imgs, labs = [], []
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        img = 127.5 + 40.0 * rng.standard_normal((P,) * rank + (cin,))
        lab = np.zeros((P,) * rank, dtype=np.int32)
        grid = np.stack(np.meshgrid(*[np.arange(P)] * rank, indexing='ij'), -1)
        for c in range(1, K):
            ctr = rng.uniform(P / 4.0, 3.0 * P / 4.0, size=rank)
            m = ((grid - ctr) ** 2).sum(-1) <= (P / 6.0) ** 2
            lab[m] = c
        img = img + 60.0 * (lab > 0)[..., None]
        imgs.append(np.clip(img, 0.0, 255.0).astype(np.float32))
        labs.append(lab[..., None])
    return np.stack(imgs), np.stack(labs).astype(np.int32)

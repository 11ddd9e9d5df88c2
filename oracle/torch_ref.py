"""Second, independent CPU restatement (TEST INFRASTRUCTURE ONLY): the same reference graph
(networks.py:246-365, VNet.py:26-155, model.py:26-85/447/474-558) wired with stock
PyTorch-CPU ops and torch.autograd.  Used (a) to pin oracle/vnet_oracle.py -- two independent
restatements agreeing to ~1e-10 in float64 -- and (b) as the timed `cpu_baseline` ("port",
oneDNN conv3d, fp32) in bench.py.  PARITY UNPINNED by the reference itself (it has no tests).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Layout mapping: TF NDHWC/DHWIO -> torch NCDHW/OIDHW via permutes; parameters
stay in TF layout and TF names so the two oracles and the HIP path share one weight dict.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-3

# Independent restatement of the bf16-storage mode of round 3 (vnet_oracle.ACT_STORAGE): STORAGE = "bf16" makes `_store` round a
# tensor to bfloat16 forward AND the gradient that flows back through it (torch's own round-to-nearest-even conversion, through
# float32 like the device), and every convolution with a spatial kernel round its filter (straight-through: the filter
# gradient, a product of bf16 tensors, goes to the fp32 master weights unrounded).
STORAGE = None


def _rb(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _StoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w):
        return _rb(w)

    @staticmethod
    def backward(ctx, g):
        return g


def _store(x):
    return _StoreFn.apply(x) if STORAGE == "bf16" else x


def _operand(t, k):
    return _RoundSTE.apply(t) if (STORAGE == "bf16" and k > 1) else t


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


def _to_ncx(x):   # [B,*S,C] -> [B,C,*S]
    r = x.dim() - 2
    return x.permute(0, r + 1, *range(1, r + 1))


def _to_nxc(x):   # [B,C,*S] -> [B,*S,C]
    r = x.dim() - 2
    return x.permute(0, *range(2, r + 2), 1)


def convolution(x, w, b, stride=1):
    """layers2.py:59-63 (tf.nn.convolution SAME) on channels-last tensors, w [*k,Ci,Co]."""
    r = x.dim() - 2
    pads = []
    for a in reversed(range(r)):            # F.pad wants last spatial dim first
        _, lo, hi = _same_pad(x.shape[1 + a], w.shape[a], stride)
        pads += [lo, hi]
    x, w = _operand(x, w.shape[0]), _operand(w, w.shape[0])
    xc = F.pad(_to_ncx(x), pads)
    wt = w.permute(r + 1, r, *range(r))     # [Co,Ci,*k]
    conv = F.conv3d if r == 3 else F.conv2d
    return _to_nxc(conv(xc, wt, None, stride)) + b


def deconvolution(x, w, b, out_spatial, stride=2):
    """layers2.py:65-74 (tf.nn.conv3d_transpose SAME), w [*k,Cout,Cin]."""
    r = x.dim() - 2
    w = _operand(w, w.shape[0])
    wt = w.permute(r + 1, r, *range(r))     # [Cin,Cout,*k] = torch conv_transpose layout
    convt = F.conv_transpose3d if r == 3 else F.conv_transpose2d
    k = w.shape[0]
    # SAME transpose: full output (in-1)*s+k, then crop pad_before of the forward conv
    y = convt(_to_ncx(x), wt, None, stride)
    crop = []
    for a in range(r):
        _, lo, _ = _same_pad(out_spatial[a], k, stride)
        crop.append(slice(lo, lo + out_spatial[a]))
    full_needed = [(x.shape[1 + a] - 1) * stride + k for a in range(r)]
    # pad on the high side if the requested output is larger than the scatter extent (odd sizes)
    padn = []
    for a in reversed(range(r)):
        _, lo, _ = _same_pad(out_spatial[a], k, stride)
        padn += [0, max(lo + out_spatial[a] - full_needed[a], 0)]
    y = F.pad(y, padn)
    y = y[(slice(None), slice(None)) + tuple(crop)]
    return _to_nxc(y) + b


def batch_norm(x, gamma, beta):
    """tf.layers.batch_normalization(training=True): biased batch statistics (A.4)."""
    ax = tuple(range(x.dim() - 1))
    mu = x.mean(dim=ax)
    var = ((x - mu) ** 2).mean(dim=ax)
    return (x - mu) * torch.rsqrt(var + BN_EPS) * gamma + beta


def prelu(x, alpha):
    """layers2.py:97-99.  torch.clamp gives subgradient 1 at 0 for both halves, TF gives 0 (A.5);
    exact zeros do not occur with continuous random inputs, the numpy oracle handles the tie."""
    zero = torch.zeros((), dtype=x.dtype)
    return torch.maximum(zero, x) + alpha * torch.minimum(zero, x)


class TorchVNet(object):
    """Functional wiring over a {tf_name: tensor} parameter dict (created lazily)."""

    def __init__(self, num_classes, num_channels=16, num_levels=4, num_convolutions=(1, 2, 3, 3),
                 bottom_convolutions=3, activation_fn="prelu", variant="networks", params=None,
                 dtype=torch.float64):
        self.K, self.C0, self.L = num_classes, num_channels, num_levels
        self.ncv, self.nb = tuple(num_convolutions), bottom_convolutions
        self.act, self.variant, self.dtype = activation_fn, variant, dtype
        self.p = params if params is not None else {}
        self.scope, self.bnc = [], {}

    # --- scoped variables ---------------------------------------------------------------
    def _name(self, n):
        return "/".join(self.scope + [n])

    def _get(self, n, shape, fill=None):
        full = self._name(n)
        if full not in self.p:
            if fill is None:     # xavier uniform, layers2.py:16-21
                s = len(shape) - 2
                num = math.prod(shape[:s]) * (shape[-2] + shape[-1])
                lim = math.sqrt(6.0 / num)
                t = (torch.rand(shape, dtype=torch.float64) * 2 - 1) * lim
            else:
                t = torch.full(shape, fill, dtype=torch.float64)
            self.p[full] = t.to(self.dtype).requires_grad_(True)
        return self.p[full]

    def _bn(self, x):
        key = "/".join(self.scope)
        n = self.bnc.get(key, 0)
        self.bnc[key] = n + 1
        self.scope.append("batch_normalization" if n == 0 else "batch_normalization_%d" % n)
        C = x.shape[-1]
        g, b = self._get('gamma', (C,), 1.0), self._get('beta', (C,), 0.0)
        self.scope.pop()
        return batch_norm(x, g, b)

    def _act(self, x):
        if self.act == 'prelu':
            return prelu(x, self._get('alpha', (x.shape[-1],), 0.1))
        if self.act == 'relu':
            return F.relu(x)
        return F.leaky_relu(x, 0.2)

    def _conv(self, x, filt, stride=1):
        return convolution(x, self._get('weights', tuple(filt)), self._get('biases', (filt[-1],), 0.0), stride)

    # --- blocks -------------------------------------------------------------------------
    def _block(self, x, n):
        inp, C, k = x, x.shape[-1], [5] * (x.dim() - 2)
        for i in range(n):
            self.scope.append('conv_%d' % (i + 1))
            x = _store(self._conv(x, k + [C, C]))
            if self.variant == 'legacy':
                x = _store(self._bn(x))
            if i == n - 1:
                x = x + inp
            x = _store(self._act(self._bn(x)))
            self.scope.pop()
        return x

    def _block2(self, x, f, n):
        inp, C, k = x, x.shape[-1], [5] * (x.dim() - 2)
        x = torch.cat((x, f), dim=-1)
        legacy = self.variant == 'legacy'
        self.scope.append('conv_1')
        x = self._bn(_store(self._conv(x, k + [2 * C, C])))
        if n == 1:
            if legacy:
                x = _store(x)
            else:
                inp = self._bn(x)
            x = _store(self._act(self._bn(x + inp)))
            self.scope.pop()
            return x
        x = _store(self._act(x))
        self.scope.pop()
        for i in range(1, n):
            self.scope.append('conv_%d' % (i + 1))
            x = _store(self._conv(x, k + [C, C]))
            if legacy:
                x = _store(self._bn(x))
            else:
                inp = self._bn(x)
            if i == n - 1:
                x = x + inp
            x = _store(self._act(self._bn(x)))
            self.scope.pop()
        return x

    def forward(self, x):
        self.bnc = {}
        r, cin = x.dim() - 2, x.shape[-1]
        self.scope = ['vnet/input_layer']
        if cin == 1:
            x = _store(self._bn(x.repeat(*([1] * (r + 1)), self.C0)))
        else:
            x = _store(self._act(self._bn(_store(self._conv(x, [5] * r + [cin, self.C0])))))
        feats = []
        for l in range(self.L):
            self.scope = ['vnet/encoder/level_%d' % (l + 1)]
            x = self._block(x, self.ncv[l])
            feats.append(x)
            self.scope.append('down_convolution')
            C = x.shape[-1]
            x = _store(self._act(self._bn(_store(self._conv(x, [2] * r + [C, 2 * C], 2)))))
        self.scope = ['vnet/bottom_level']
        x = self._block(x, self.nb)
        for l in reversed(range(self.L)):
            self.scope = ['vnet/decoder/level_%d' % (l + 1), 'up_convolution']
            f, C = feats[l], x.shape[-1]
            w = self._get('weights', tuple([2] * r + [C // 2, C]))
            b = self._get('biases', (C // 2,), 0.0)
            x = _store(self._act(self._bn(_store(deconvolution(x, w, b, f.shape[1:-1], 2)))))
            self.scope.pop()
            x = self._block2(x, f, self.ncv[l])
        self.scope = ['vnet/output_layer']
        return self._bn(self._conv(x, [1] * r + [self.C0, self.K]))


def dice_coe(output, target, loss_type='jaccard', axis=(1, 2, 3), weights=(), smooth=1e-5):
    """model.py:26-85."""
    inse = (output * target).sum(dim=axis)
    if loss_type == 'jaccard':
        l, r = (output * output).sum(dim=axis), (target * target).sum(dim=axis)
    elif loss_type == 'sorensen':
        l, r = output.sum(dim=axis), target.sum(dim=axis)
    else:
        raise Exception("Unknown loss_type")
    if len(weights) != 0:
        w = torch.as_tensor(weights, dtype=output.dtype)
        dice = (2. * w * inse + smooth).sum(-1) / (w * (l + r) + smooth).sum(-1)
    else:
        dice = (2. * inse + smooth) / (l + r + smooth)
    return dice.mean()


def loss_head(logits, labels, loss_name='sorensen', weights=(), alpha=1.0):
    """model.py:447,474-558.  labels int [B,*S,1]."""
    K = logits.shape[-1]
    lab = labels[..., 0].long()
    oh = ((lab.unsqueeze(-1) == torch.arange(K)).to(logits.dtype))
    sm = torch.softmax(logits, dim=-1)
    axis = tuple(range(1, logits.dim() - 1))

    def xent(wts=None):
        per = -(oh * torch.log_softmax(logits, -1)).sum(-1)
        if wts is not None:
            per = per * (torch.as_tensor(wts, dtype=logits.dtype) * oh).sum(-1)
        return per.mean()
    if loss_name == 'xent':
        return xent(), sm
    if loss_name == 'weighted_xent':
        return xent(weights), sm
    kind = 'sorensen' if 'sorensen' in loss_name else 'jaccard'
    w = weights if 'weighted' in loss_name else ()
    loss = 1.0 - dice_coe(sm, oh, kind, axis, w)
    if loss_name.startswith('mixed_'):
        loss = loss + alpha * xent(weights if 'weighted' in loss_name else None)
    return loss, sm


def train_step_fp32(net, images, labels, opt):
    """One fp32 CPU training step (fwd + Dice + bwd + optimiser) -- the timed cpu_baseline."""
    opt.zero_grad(set_to_none=True)
    loss, _ = loss_head(net.forward(images), labels, 'sorensen')
    loss.backward()
    opt.step()
    return float(loss.detach())

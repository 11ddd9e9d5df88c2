"""Minimal stand-in for tf.variable_scope / tf.get_variable / tf.layers name uniquification, so
the mirrored layers2 / networks / VNet modules create torch Parameters under exactly the TF
variable names of the reference (e.g. 'vnet/encoder/level_1/conv_1/weights')."""
import contextlib
from collections import OrderedDict

import numpy as np
import torch

_ACTIVE = []


class VariableStore(object):
    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else None
        self.params = OrderedDict()    # trainable (tf.trainable_variables)
        self.buffers = OrderedDict()   # non-trainable (BN moving statistics)
        self._scope = []
        self._bn_count = {}
        self.values = None             # optional {name: array} injected instead of initialisers

    # -- graph-pass bookkeeping --------------------------------------------------------------
    def begin_pass(self):
        self._scope = []
        self._bn_count = {}

    @contextlib.contextmanager
    def active(self):
        _ACTIVE.append(self)
        try:
            yield self
        finally:
            _ACTIVE.pop()

    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name):
        return "/".join(self._scope + [name])

    def unique_layer_name(self, base):
        """tf.layers auto-names: base, base_1, base_2 ... per enclosing variable scope."""
        key = ("/".join(self._scope), base)
        n = self._bn_count.get(key, 0)
        self._bn_count[key] = n + 1
        return base if n == 0 else "%s_%d" % (base, n)

    # -- variables ---------------------------------------------------------------------------
    def _materialise(self, full, init):
        if self.values is not None and full in self.values:
            v = self.values[full]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        else:
            v = init()
        t = torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32)
        return t.to(self.device) if self.device is not None else t

    def get_variable(self, name, initializer, trainable=True):
        full = self.full_name(name)
        table = self.params if trainable else self.buffers
        if full not in table:
            t = self._materialise(full, initializer if callable(initializer) else (lambda: initializer))
            table[full] = torch.nn.Parameter(t, requires_grad=True) if trainable else t
        return table[full]

    # -- (de)serialisation -------------------------------------------------------------------
    def state_dict(self):
        out = OrderedDict()
        for k, v in self.params.items():
            out[k] = v.detach()
        for k, v in self.buffers.items():
            out[k] = v
        return out

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, v in sd.items():
                dst = self.params.get(k, self.buffers.get(k))
                if dst is None:
                    raise KeyError("unknown variable %r" % k)
                dst.copy_(torch.as_tensor(v).to(dst.device, dst.dtype).reshape(dst.shape))


def current():
    if not _ACTIVE:
        raise RuntimeError("no active VariableStore: call inside `with store.active():` (the stand-in for a tf graph)")
    return _ACTIVE[-1]


def variable_scope(name):
    return current().variable_scope(name)


def get_variable(name, initializer=None, shape=None, trainable=True):
    if initializer is None:
        raise ValueError("initializer required")
    return current().get_variable(name, initializer, trainable)

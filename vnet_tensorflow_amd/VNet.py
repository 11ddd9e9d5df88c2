"""Mirror of the reference's legacy VNet.py (VNet.py:26-155, used only by the stale train.py):
two batch-norms per convolution with the residual added between them (VNet.py:31-38), the
decoder residual is the real up-convolution output (VNet.py:50,68), dropout by `keep_prob`.
Same kernels as networks.VNet, different wiring; 3-D only like the reference."""
import torch

from . import layers2 as L
from . import ops
from ._scope import VariableStore, current


def convolution_block(layer_input, num_convolutions, keep_prob, activation_fn, is_training=True, tiled=None):
    """reference VNet.py:26-39"""
    store = current()
    x = layer_input
    n_channels = L.get_num_channels(x)
    for i in range(num_convolutions):
        with store.variable_scope('conv_' + str(i + 1)):
            if i == 0 and tiled is not None:
                x = L.convolution_tiled(tiled, [5, 5, 5, n_channels, n_channels])
            else:
                x = L.convolution(x, [5, 5, 5, n_channels, n_channels])
            x = L.batch_normalization(x)
            res = layer_input if i == num_convolutions - 1 else None
            x = L.batch_normalization(x, activation=activation_fn, residual=res)
            x = ops.dropout(x, 1.0 - keep_prob)
    return x


def convolution_block_2(layer_input, fine_grained_features, num_convolutions, keep_prob, activation_fn, is_training=True):
    """reference VNet.py:42-73"""
    store = current()
    n_channels = L.get_num_channels(layer_input)
    if num_convolutions == 1:
        with store.variable_scope('conv_' + str(1)):
            x = L.convolution_concat(layer_input, fine_grained_features, [5, 5, 5, n_channels * 2, n_channels])
            x = L.batch_normalization(x)
            x = L.batch_normalization(x, activation=activation_fn, residual=layer_input)
            x = ops.dropout(x, 1.0 - keep_prob)
        return x

    with store.variable_scope('conv_' + str(1)):
        x = L.convolution_concat(layer_input, fine_grained_features, [5, 5, 5, n_channels * 2, n_channels])
        x = L.batch_normalization(x, activation=activation_fn)
        x = ops.dropout(x, 1.0 - keep_prob)

    for i in range(1, num_convolutions):
        with store.variable_scope('conv_' + str(i + 1)):
            x = L.convolution(x, [5, 5, 5, n_channels, n_channels])
            x = L.batch_normalization(x)
            res = layer_input if i == num_convolutions - 1 else None
            x = L.batch_normalization(x, activation=activation_fn, residual=res)
            x = ops.dropout(x, 1.0 - keep_prob)
    return x


class VNet(object):
    def __init__(self,
                 num_classes,
                 keep_prob=1.0,
                 num_channels=16,
                 num_levels=4,
                 num_convolutions=(1, 2, 3, 3),
                 bottom_convolutions=3,
                 is_training=True,
                 activation_fn="relu",
                 device=None):
        """reference VNet.py:76-108"""
        self.num_classes = num_classes
        self.keep_prob = keep_prob
        self.num_channels = num_channels
        assert num_levels == len(num_convolutions)
        self.num_levels = num_levels
        self.num_convolutions = num_convolutions
        self.bottom_convolutions = bottom_convolutions
        self.is_training = is_training
        self.train_phase = True
        if activation_fn not in ("relu", "prelu"):
            raise ValueError("activation_fn must be relu or prelu")
        self.activation_fn = activation_fn
        self.fuse_input_block = True
        self.fuse_zero_bias_grad = True    # every conv feeds a batch-norm (VNet.py:32,36): bias gradients are identically 0
        self.variables = VariableStore(device)

    def parameters(self):
        return list(self.variables.params.values())

    def named_parameters(self):
        return list(self.variables.params.items())

    def state_dict(self):
        return self.variables.state_dict()

    def load_state_dict(self, sd):
        self.variables.load_state_dict(sd)

    def build(self, input_shape):
        with torch.no_grad():
            self.network_fn(torch.empty(tuple(input_shape), device="meta"))
        return self

    def network_fn(self, x):
        """reference VNet.py:110-155"""
        store = self.variables
        if store.device is None and x.device.type != "meta":
            store.device = x.device
        store.begin_pass()
        keep_prob = float(self.keep_prob() if callable(self.keep_prob) else self.keep_prob)
        act = self.activation_fn
        from . import ops
        with store.active(), ops.zero_bias_gradients(self.fuse_zero_bias_grad):
            input_channels = int(x.shape[-1])
            with store.variable_scope('vnet/input_layer'):
                tiled = None
                store16 = ops.storage_is_bf16() and x.device.type != "meta"
                if store16 and (self.num_channels < 8 or self.num_channels & (self.num_channels - 1)):
                    raise ops.VnetHipError("ComputeDtype 'bf16' stores activations as bf16 in 16-byte channel units: NumChannel must be "
                                           "8 * 2^k (got %d); use 'fp32' or 'fp32_split3'" % self.num_channels)
                if input_channels == 1:
                    # tile + BN; the first 5^3 conv then runs on the un-tiled image (layers2.convolution_tiled)
                    x, tiled = L.batch_normalization(x, tile=True, channels=self.num_channels, want_stats=True)
                    if self.num_channels > 16 or not self.fuse_input_block or store16:
                        tiled = None
                else:
                    if store16:
                        x = ops.cast_input(x)     # bf16 storage mode: see networks.VNet.GetNetwork
                    x = L.convolution(x, [5, 5, 5, input_channels, self.num_channels])
                    x = L.batch_normalization(x, activation=act)

            features = list()
            for l in range(self.num_levels):
                with store.variable_scope('vnet/encoder/level_' + str(l + 1)):
                    x = convolution_block(x, self.num_convolutions[l], keep_prob, activation_fn=act,
                                          tiled=tiled if l == 0 else None)
                    features.append(x)
                    with store.variable_scope('down_convolution'):
                        x = L.down_convolution(x, factor=2, kernel_size=[2, 2, 2])
                        x = L.batch_normalization(x, activation=act)

            with store.variable_scope('vnet/bottom_level'):
                x = convolution_block(x, self.bottom_convolutions, keep_prob, activation_fn=act)

            for l in reversed(range(self.num_levels)):
                with store.variable_scope('vnet/decoder/level_' + str(l + 1)):
                    f = features[l]
                    with store.variable_scope('up_convolution'):
                        x = L.up_convolution(x, tuple(f.shape), factor=2, kernel_size=[2, 2, 2])
                        x = L.batch_normalization(x, activation=act)
                    x = convolution_block_2(x, f, self.num_convolutions[l], keep_prob, activation_fn=act)

            with store.variable_scope('vnet/output_layer'):
                logits = L.convolution(x, [1, 1, 1, self.num_channels, self.num_classes])
                logits = L.batch_normalization(logits)
        return logits

"""Mirror of the reference's networks.py `class VNet` (networks.py:209-365) -- the network main.py
uses (model.py:428-438) -- wired onto the fused HIP ops.

Same constructor signature / defaults, same `GetNetwork(x)` entry point, same TF variable names
(SURVEY.md B.1).  Differences that do not change results: the channel concat (networks.py:325)
is never materialised (the conv reads two sources), tf.tile of the 1-channel input
(networks.py:258), the residual adds (networks.py:318,336,360) and the activation are fused
into the batch-norm kernels.  `train_phase` is kept as an attribute for API parity; like the
reference (which feeds True in train, test and evaluate, model.py:747,788,917) batch statistics
are always used.
"""
import torch

from . import layers2 as L
from ._scope import VariableStore


class VNet(object):
    def __init__(self,
                 num_classes,
                 dropout_rate=0.01,
                 num_channels=16,
                 num_levels=4,
                 num_convolutions=(1, 2, 3, 3),
                 bottom_convolutions=3,
                 is_training=True,
                 activation_fn="relu",
                 device=None):
        """Implements VNet architecture https://arxiv.org/abs/1606.04797 (reference networks.py:210-244)."""
        self.num_classes = num_classes
        self.dropout_rate = dropout_rate
        self.num_channels = num_channels
        assert num_levels == len(num_convolutions)
        self.num_levels = num_levels
        self.num_convolutions = num_convolutions
        self.bottom_convolutions = bottom_convolutions
        self.is_training = is_training
        self.train_phase = True            # stand-in for the "train_phase_placeholder" (networks.py:237)
        if activation_fn not in ("relu", "prelu", "lrelu"):
            raise ValueError("activation_fn must be relu, prelu or lrelu")
        self.activation_fn = activation_fn
        self.fuse_input_block = True       # single-modality input: skip the 16x redundant work of the tiled conv
        self.fuse_bn_chains = True         # decoder BN->BN->add->BN chains in closed form (ops.bn_chain)
        self.cut_backward = False          # data-parallel step graphs: cut the autograd graph between encoder and bottom level / skips
        self.backward_cuts = []
        self.fuse_bn_stats = True          # batch-norm statistics from the producing convolution's epilogue (ops.conv bn_stats)
        self.fuse_grad_accumulation = True # tensors with two consumers: second gradient accumulated by its producer (ops.fork)
        self.fuse_zero_bias_grad = True    # conv biases feed batch-norms: their gradient is identically 0 (ops.zero_bias_gradients)
        self.variables = VariableStore(device)

    # -- torch.nn.Module-like conveniences -------------------------------------------------
    def parameters(self):
        return list(self.variables.params.values())

    def named_parameters(self):
        return list(self.variables.params.items())

    def state_dict(self):
        return self.variables.state_dict()

    def load_state_dict(self, sd):
        self.variables.load_state_dict(sd)

    def build(self, input_shape):
        """Create all variables (on self.variables.device) from a shape only -- replaces the
        graph-construction pass of TF (model.py:444)."""
        with torch.no_grad():
            self.GetNetwork(torch.empty(tuple(input_shape), device="meta"))
        return self

    def _dropout_rate(self):
        r = self.dropout_rate if self.is_training else 0.0
        return float(r() if callable(r) else r)

    # -- reference networks.py:246-305 ---------------------------------------------------------
    def GetNetwork(self, x):
        store = self.variables
        if store.device is None and x.device.type != "meta":
            store.device = x.device
        store.begin_pass()
        dropout_rate = self._dropout_rate()
        act = self.activation_fn
        from . import ops
        with store.active(), ops.zero_bias_gradients(self.fuse_zero_bias_grad):
            if x.dim() != 5:
                raise NotImplementedError("only 3-D PatchShape is built (2-D is out of scope, SURVEY section 2 row 11)")
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
                        tiled = None          # (bf16 storage: the tiled batch-norm writes a bf16 tensor, conv_1 is an ordinary bf16 conv)
                else:
                    if store16:
                        x = ops.cast_input(x)     # bf16, channels zero-padded to the 16-byte unit; from here on every tensor is bf16
                    x = L.convolution(x, [5, 5, 5, input_channels, self.num_channels], bn_stats=self.fuse_bn_stats)
                    x = L.batch_normalization(x, activation=act)

            features = list()
            cutting = self.cut_backward and x.device.type != "meta" and torch.is_grad_enabled()
            self.backward_cuts = []
            for l in range(self.num_levels):
                with store.variable_scope('vnet/encoder/level_' + str(l + 1)):
                    x = self.convolution_block(x, self.num_convolutions[l], dropout_rate, act,
                                               tiled=tiled if l == 0 else None)
                    if self.fuse_grad_accumulation and not cutting:
                        # skip connection: decoder concat + down convolution consume x (ops.fork)
                        skip, x = ops.fork(x)
                        features.append(skip)
                    else:
                        features.append(x)
                    with store.variable_scope('down_convolution'):
                        x = L.down_convolution(x, factor=2, kernel_size=[2, 2, 2], bn_stats=self.fuse_bn_stats)
                        x = L.batch_normalization(x, activation=act)

            if cutting:
                # everything the decoder and the bottom level take from the encoder becomes a leaf: backward pass 1 ends here
                features = [ops.cut(f, self.backward_cuts) for f in features]
                x = ops.cut(x, self.backward_cuts)
            with store.variable_scope('vnet/bottom_level'):
                x = self.convolution_block(x, self.bottom_convolutions, dropout_rate, act)

            for l in reversed(range(self.num_levels)):
                with store.variable_scope('vnet/decoder/level_' + str(l + 1)):
                    f = features[l]
                    with store.variable_scope('up_convolution'):
                        x = L.up_convolution(x, tuple(f.shape), factor=2, kernel_size=[2, 2, 2])
                        x = L.batch_normalization(x, activation=act)
                    x = self.convolution_block_2(x, f, self.num_convolutions[l], dropout_rate, act)

            with store.variable_scope('vnet/output_layer'):
                logits = L.convolution(x, [1, 1, 1, self.num_channels, self.num_classes])
                logits = L.batch_normalization(logits)
        return logits

    # -- reference networks.py:307-322 ---------------------------------------------------------
    def convolution_block(self, layer_input, num_convolutions, dropout_rate, activation_fn, is_training=True, tiled=None):
        from . import ops
        store = self.variables
        x = layer_input
        n_channels = L.get_num_channels(x)
        if tiled is None and self.fuse_grad_accumulation:
            # the block input feeds conv_1 and the residual add: its second gradient is accumulated in place (ops.fork)
            x, layer_input = ops.fork(layer_input)
        for i in range(num_convolutions):
            with store.variable_scope('conv_' + str(i + 1)):
                res = layer_input if i == num_convolutions - 1 else None          # x = x + layer_input
                if i == 0 and tiled is not None:
                    x = L.convolution_tiled(tiled, [5, 5, 5, n_channels, n_channels], bn_stats=self.fuse_bn_stats, bn_residual=res)
                else:
                    x = L.convolution(x, [5, 5, 5, n_channels, n_channels], bn_stats=self.fuse_bn_stats, bn_residual=res)
                x = L.batch_normalization(x, activation=activation_fn, residual=res)
                x = ops.dropout(x, dropout_rate)
        return x

    # -- reference networks.py:324-365 ---------------------------------------------------------
    def convolution_block_2(self, layer_input, fine_grained_features, num_convolutions, dropout_rate, activation_fn,
                            is_training=True):
        from . import ops
        store = self.variables
        n_channels = L.get_num_channels(layer_input)
        if num_convolutions == 1:
            with store.variable_scope('conv_' + str(1)):
                x = L.convolution_concat(layer_input, fine_grained_features, [5, 5, 5, n_channels * 2, n_channels], bn_stats=self.fuse_bn_stats)
                if self.fuse_bn_chains:
                    # x = BN(x); r = BN(x) (networks.py:335); x = act(BN(x + r)) -- one fused normalisation of the conv output
                    x = L.batch_normalization_chain(x, 0, activation_fn)
                else:
                    x = L.batch_normalization(x)
                    r = L.batch_normalization(x)                                       # networks.py:335
                    x = L.batch_normalization(x, activation=activation_fn, residual=r)  # x = x + layer_input ; BN ; act
                x = ops.dropout(x, dropout_rate)
            return x

        with store.variable_scope('conv_' + str(1)):
            x = L.convolution_concat(layer_input, fine_grained_features, [5, 5, 5, n_channels * 2, n_channels], bn_stats=self.fuse_bn_stats)
            x = L.batch_normalization(x, activation=activation_fn)
            x = ops.dropout(x, dropout_rate)

        for i in range(1, num_convolutions):
            with store.variable_scope('conv_' + str(i + 1)):
                x = L.convolution(x, [5, 5, 5, n_channels, n_channels], bn_stats=self.fuse_bn_stats)
                last = (i == num_convolutions - 1)
                # networks.py:358 builds this BN for every i; its output is used only by the last conv
                if last and self.fuse_bn_chains:
                    x = L.batch_normalization_chain(x, 1, activation_fn)               # r = BN(x); x = act(BN(x + r))
                else:
                    r = L.batch_normalization(x, dead=not last)
                    x = L.batch_normalization(x, activation=activation_fn, residual=r if last else None)
                x = ops.dropout(x, dropout_rate)
        return x

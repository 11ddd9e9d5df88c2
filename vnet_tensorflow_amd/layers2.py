"""Mirror of the reference's layers2.py (same function names, argument order and meaning) on top of
the HIP library.  Tensors are NDHWC torch tensors on a HIP device; variables are created through
the scoped store of `_scope.py` under the TF names ('weights', 'biases', 'alpha').

reference layers2.py:4-30   xavier_initializer_convolution
reference layers2.py:32-57  constant_initializer / get_num_channels / get_spatial_rank / get_spatial_size
reference layers2.py:59-63  convolution
reference layers2.py:65-74  deconvolution
reference layers2.py:78-94  down_convolution / up_convolution
reference layers2.py:97-99  prelu
"""
import numpy as np

from . import ops
from ._scope import get_variable, variable_scope, current


def xavier_initializer_convolution(shape, dist='uniform', lambda_initializer=True):
    """Xavier initialiser for N-D convolution patches (reference layers2.py:4-30; note the code's
    sqrt(6/n) for uniform, not the docstring's sqrt(3/n)).  Uses the global NumPy RNG like the reference."""
    s = len(shape) - 2
    num_activations = np.prod(shape[:s]) * np.sum(shape[s:])
    if dist == 'uniform':
        lim = np.sqrt(6. / num_activations)
        return np.random.uniform(-lim, lim, shape).astype(np.float32)
    if dist == 'normal':
        stddev = np.sqrt(3. / num_activations)
        return np.random.normal(0, stddev, shape).astype(np.float32)
    raise ValueError('Distribution must be either "uniform" or "normal".')


def constant_initializer(value, shape, lambda_initializer=True):
    return np.full(shape, value).astype(np.float32)


def get_num_channels(x):
    return int(x.shape[-1])


def get_spatial_rank(x):
    return x.dim() - 2


def get_spatial_size(x):
    return tuple(x.shape[1:-1])


def _check_filter(filter, rank):
    if rank != 3:
        raise NotImplementedError("only 3-D volumes are built (2-D PatchShape is out of scope, SURVEY section 2 row 11)")
    k = list(filter[:rank])
    if len(set(k)) != 1:
        raise ValueError("anisotropic kernels are not supported")
    return k[0]


def convolution(x, filter, padding='SAME', strides=None, dilation_rate=None, initializer="XAVIER", bn_stats=False, bn_residual=None):
    """tf.nn.convolution(x, w, padding, strides, dilation_rate) + b with variables 'weights'/'biases'."""
    if padding != 'SAME':
        raise ValueError("only SAME padding is used by the reference networks")
    if dilation_rate not in (None, 1) and any(d != 1 for d in np.atleast_1d(dilation_rate)):
        raise ValueError("dilation is not used by the reference networks")
    filter = list(filter)
    w = get_variable(name='weights', initializer=lambda: xavier_initializer_convolution(shape=filter))
    b = get_variable(name='biases', initializer=lambda: constant_initializer(0, shape=filter[-1]))
    rank = get_spatial_rank(x)
    k = _check_filter(filter, rank)
    s = 1 if strides is None else int(np.atleast_1d(strides)[0])
    if k == 1 and s == 1:
        return ops.head_conv(x, w, b)
    if not ((k == 5 and s == 1) or (k == 2 and s == 2)):
        raise NotImplementedError("kernel %d stride %d is not instantiated (V-Net uses 5/1, 2/2, 1/1)" % (k, s))
    # bn_stats / bn_residual (extension): the caller normalises this output (+ residual) next -- see ops.conv
    return ops.conv(x, w, b, k, s, bn_stats=bn_stats, bn_residual=bn_residual)


def convolution_concat(x, skip, filter, bn_stats=False):
    """convolution(tf.concat((x, skip), -1), filter) without materialising the concat (networks.py:325)."""
    filter = list(filter)
    w = get_variable(name='weights', initializer=lambda: xavier_initializer_convolution(shape=filter))
    b = get_variable(name='biases', initializer=lambda: constant_initializer(0, shape=filter[-1]))
    k = _check_filter(filter, get_spatial_rank(x))
    return ops.conv(x, w, b, k, 1, x1=skip, bn_stats=bn_stats)


def deconvolution(x, filter, output_shape, strides, padding='SAME'):
    """tf.nn.conv3d_transpose(x, w, output_shape, strides, padding) + b; filter = k + [Cout, Cin];
    the bias has filter[-2] elements (reference layers2.py:67)."""
    filter = list(filter)
    w = get_variable(name='weights', initializer=lambda: xavier_initializer_convolution(shape=filter))
    b = get_variable(name='biases', initializer=lambda: constant_initializer(0, shape=filter[-2]))
    rank = get_spatial_rank(x)
    k = _check_filter(filter, rank)
    s = list(strides)
    if k != 2 or s != [1] + rank * [2] + [1]:
        raise NotImplementedError("only the 2x2x2 stride-2 transposed convolution of the V-Net is instantiated")
    out_spatial = tuple(int(v) for v in tuple(output_shape)[1:-1])
    return ops.conv_transpose2(x, w, b, out_spatial)


def down_convolution(x, factor, kernel_size, bn_stats=False):
    num_channels = get_num_channels(x)
    spatial_rank = get_spatial_rank(x)
    strides = spatial_rank * [factor]
    filter = list(kernel_size) + [num_channels, num_channels * factor]
    return convolution(x, filter, strides=strides, bn_stats=bn_stats)


def up_convolution(x, output_shape, factor, kernel_size):
    num_channels = get_num_channels(x)
    spatial_rank = get_spatial_rank(x)
    strides = [1] + spatial_rank * [factor] + [1]
    filter = list(kernel_size) + [num_channels // factor, num_channels]
    return deconvolution(x, filter, output_shape, strides=strides)


def prelu_alpha(x):
    return get_variable('alpha', initializer=lambda: np.full((int(x.shape[-1]),), 0.1, dtype=np.float32))


def prelu(x):
    """tf.maximum(0.0, x) + alpha * tf.minimum(0.0, x), alpha initialised to 0.1 (reference layers2.py:97-99).
    Stand-alone form; inside the networks the activation is fused into batch_normalization()."""
    return ops.activation(x, "prelu", prelu_alpha(x))


def relu(x):
    return ops.activation(x, "relu")


def leaky_relu(x):
    return ops.activation(x, "lrelu")


def convolution_tiled(tiled, filter, bn_stats=False, bn_residual=None):
    """convolution(x, filter) where x = batch_normalization(tf.tile(img)) of a 1-channel image (reference
    networks.py:254-259 followed by networks.py:316): same variables ('weights', 'biases'), same result up to fp32
    summation order, 5x fewer matrix instructions forward / in the filter gradient and no backward-data pass
    (csrc/input_block.hip).  `tiled` = (img, gamma, beta, mean, invstd) from batch_normalization(..., want_stats=True)."""
    filter = list(filter)
    w = get_variable(name='weights', initializer=lambda: xavier_initializer_convolution(shape=filter))
    b = get_variable(name='biases', initializer=lambda: constant_initializer(0, shape=filter[-1]))
    img, gamma, beta, mean, invstd = tiled
    return ops.input_conv(img, gamma, beta, mean, invstd, w, b, bn_stats=bn_stats, bn_residual=bn_residual)


def batch_normalization(x, activation=None, residual=None, tile=False, channels=None, dead=False,
                        momentum=0.99, epsilon=0.001, want_stats=False):
    """tf.layers.batch_normalization(x, momentum=0.99, epsilon=0.001, center=True, scale=True,
    training=True) -- the reference feeds train_phase=True everywhere (model.py:747,788,917) --
    fused with the optional residual add in front (x + residual), the tf.tile of a 1-channel
    input (tile=True) and the activation behind it.  Variables live under the auto-uniquified
    layer scope 'batch_normalization[_N]'; 'alpha' lives in the enclosing scope like the reference."""
    store = current()
    C = int(channels if channels is not None else x.shape[-1])
    with variable_scope(store.unique_layer_name("batch_normalization")):
        gamma = get_variable('gamma', initializer=lambda: np.ones((C,), np.float32))
        beta = get_variable('beta', initializer=lambda: np.zeros((C,), np.float32))
        mm = get_variable('moving_mean', initializer=lambda: np.zeros((C,), np.float32), trainable=False)
        mv = get_variable('moving_variance', initializer=lambda: np.ones((C,), np.float32), trainable=False)
    if dead:
        ops.bn_update_only(x, C, mm, mv)
        return None
    alpha = None
    if activation == "prelu":
        alpha = get_variable('alpha', initializer=lambda: np.full((C,), 0.1, dtype=np.float32))
    if want_stats:
        y, mean, invstd = ops.bn_act(x, gamma, beta, activation, alpha, residual, tile, mm, mv, want_stats=True)
        return y, (x, gamma, beta, mean, invstd)
    return ops.bn_act(x, gamma, beta, activation, alpha, residual, tile, mm, mv)


def _bn_variables(C):
    """gamma, beta, moving_mean, moving_variance of the next auto-uniquified 'batch_normalization[_N]' layer scope."""
    store = current()
    with variable_scope(store.unique_layer_name("batch_normalization")):
        gamma = get_variable('gamma', initializer=lambda: np.ones((C,), np.float32))
        beta = get_variable('beta', initializer=lambda: np.zeros((C,), np.float32))
        mm = get_variable('moving_mean', initializer=lambda: np.zeros((C,), np.float32), trainable=False)
        mv = get_variable('moving_variance', initializer=lambda: np.ones((C,), np.float32), trainable=False)
    return gamma, beta, mm, mv


def batch_normalization_chain(x, kind, activation=None):
    """The decoder's batch-norm chains (reference networks.py:333-337 / 358-361), each evaluated as ONE fused
    normalisation of x (see ops.bn_chain): same variables, created in the reference's order and under its names,
        kind 0:  x = BN(x); r = BN(x); x = act(BN(x + r))          kind 1:  r = BN(x); x = act(BN(x + r))."""
    C = int(x.shape[-1])
    layers = [_bn_variables(C) for _ in range(3 if kind == 0 else 2)]
    alpha = None
    if activation == "prelu":
        alpha = get_variable('alpha', initializer=lambda: np.full((C,), 0.1, dtype=np.float32))
    (g1, b1, mm1, mv1), (g2, b2, mm2, mv2) = layers[0], layers[1]
    g3 = b3 = mm3 = mv3 = None
    if kind == 0:
        g3, b3, mm3, mv3 = layers[2]
    return ops.bn_chain(x, kind, activation, alpha, g1, b1, g2, b2, g3, b3, (mm1, mv1, mm2, mv2, mm3, mv3))

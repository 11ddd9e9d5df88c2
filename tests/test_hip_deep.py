"""-m gpu: the deep-level kernel of the bf16-storage 5^3 convolution (csrc/conv_deep.h; vnet_conv_fwd_b16 takes it for few bricks
and whole 32-cout blocks / 16-cin chunks: 32^3 64->64, 16^3 128->128, 8^3 256->256 and their two-source / backward-data relatives;
reference call sites layers2.py:59-63 from networks.py:280-282,307-322 and their gradients, model.py:660).

Bar: the stored bf16 value is a correct rounding of the exact (fp64 oracle) result on the same bf16-valued inputs and >= 99.5 % of
the values equal RNE(exact) outright -- the per-kernel bar of tests/test_hip_b16.py -- for forward and backward-data (the same
kernel on the transposed filter), with and without the K split over workgroups, through the ring of three tile buffers (more than
three chunks per workgroup), with ragged bricks, two sources, two destinations, accumulate mode and the epilogue statistics; and
the result must not depend on anything but the inputs (two launches: equal bits)."""
import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.test_hip_b16 import rb, g16, check_bf16, _conv5_inputs, BF
from tests.util import g, check_close

pytestmark = pytest.mark.gpu

#        B, D,  H,  W,  C0, C1, Cout                what it exercises
DEEP_SHAPES = [
    (1, 8, 8, 8, 64, 0, 64),        # 2 bricks x 2 cout blocks: K split over workgroups (one chunk each), fp32 slabs + reduce
    (1, 16, 16, 16, 32, 32, 64),    # two sources (decoder concat), chunks never straddle them
    (1, 8, 8, 8, 256, 0, 256),      # the bottom level itself: 16 slabs
    (2, 5, 9, 17, 32, 0, 32),       # ragged everything, batch 2: masked halo and masked epilogue
    (1, 6, 10, 12, 48, 16, 96),     # three cout blocks, sources of 3 + 1 chunks
    (1, 32, 32, 32, 64, 0, 64),     # level 2 itself: 256 workgroups, no split, four chunks through the three-buffer ring
    (1, 16, 16, 16, 128, 0, 128),   # level 3 itself
]


@pytest.mark.parametrize("target", [None, "1"])
@pytest.mark.parametrize("shape", DEEP_SHAPES)
def test_deep_conv_forward_backward_against_oracle(dev, shape, target, monkeypatch, lib_option):
    """target "1": no K split over workgroups -- every chunk of the layer in ONE workgroup (up to 16: the ring wraps five times)."""
    from vnet_tensorflow_amd import ops
    if target is not None:
        lib_option("BF16_DEEP_TARGET", target)
    B, D, H, W, C0, C1, Co = shape
    if target is not None and B * D * H * W * (C0 + C1) * Co > 5e8:
        pytest.skip("one workgroup per brick at this size only repeats the default plan")
    x0, x1, w, b, dy = _conv5_inputs(shape, sum(shape) + 41)
    xcat = x0 if x1 is None else np.concatenate((x0, x1), -1)
    y_ex = O.conv_nd_fwd(xcat, rb(w), 1) + b
    dx_ex, dw_ex = O.conv_nd_bwd(xcat, rb(w), dy, 1)
    outs = {}
    for deep in ("1", "0"):
        lib_option("BF16_DEEP", deep)
        tx0 = g16(x0, dev).requires_grad_(True)
        tx1 = g16(x1, dev).requires_grad_(True) if C1 else None
        tw, tb = g(w, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
        y = ops.conv(tx0, tw, tb, 5, 1, x1=tx1)
        tag = "deep=%s %s" % (deep, shape)
        check_bf16(tag + " fwd", y, y_ex)
        y.backward(g16(dy, dev))
        check_bf16(tag + " dx0", tx0.grad, dx_ex[..., :C0])
        if C1:
            check_bf16(tag + " dx1", tx1.grad, dx_ex[..., C0:])
        check_close(tag + " dw", tw.grad, dw_ex, 2e-6)
        # the bias gradient: a column sum of dy (vnet_colsum_b16: any channel count that is a multiple of 8, e.g. 96)
        check_close(tag + " db", tb.grad, dy.reshape(-1, Co).sum(0), 2e-6, atol=1e-6 * float(np.abs(dy).reshape(-1, Co).sum(0).max()))
        outs[deep] = (y.detach().clone(), tx0.grad.clone())
    # the two kernels sum in different orders: equal up to a rounding flip here and there, and NOT equal everywhere (else the
    # deep kernel did not run)
    same = (outs["1"][0] == outs["0"][0]).float().mean().item()
    assert same > 0.99, same
    assert same < 1.0 or (C0 + C1) * 125 <= 2000, "identical bits: the deep kernel did not run"


@pytest.mark.parametrize("shape", [(1, 8, 8, 8, 64, 0, 64), (1, 32, 32, 32, 64, 0, 64), (2, 5, 9, 17, 32, 0, 32), (1, 8, 16, 16, 64, 0, 32)])
@pytest.mark.parametrize("target", [None, "1"])
def test_deep_conv_statistics_accumulate_and_determinism(dev, shape, target, monkeypatch, lib_option):
    """Epilogue statistics (of the ROUNDED output + residual; from the kernel's own epilogue without a K split, from the reduce
    kernel with one), accumulate mode in place and out of place, two launches bit-equal."""
    from vnet_tensorflow_amd import ops
    if target is not None:
        lib_option("BF16_DEEP_TARGET", target)
    B, D, H, W, C0, C1, Co = shape
    x0, _, w, b, dy = _conv5_inputs(shape, sum(shape) + 7)
    rng = np.random.default_rng(3)
    res = rb(rng.standard_normal((B, D, H, W, Co)))
    tx, tw, tb, tres = g16(x0, dev), g(w, dev), g(b, dev), g16(res, dev)
    y = ops.conv(tx, tw, tb, 5, 1, bn_stats=True, bn_residual=tres)
    st = getattr(y, "_vnet_stats", None)
    assert st is not None, "this shape should produce epilogue statistics"
    part = st.partial.double().sum(0).cpu().numpy()
    v = (y.detach().float() + tres.float()).double().reshape(-1, Co).cpu().numpy()
    assert np.allclose(part[:Co], v.sum(0), rtol=2e-5, atol=1e-6 * np.abs(v).sum(0).max())
    assert np.allclose(part[Co:], (v * v).sum(0), rtol=2e-5)
    check_bf16("stats launch fwd", y, O.conv_nd_fwd(x0, rb(w), 1) + b)
    y2 = ops.conv(tx, tw, tb, 5, 1, bn_stats=True, bn_residual=tres)
    assert torch.equal(y2.detach(), y.detach()) and torch.equal(y2._vnet_stats.partial, st.partial)
    wp = ops.packed_weights(tw, ops.PACK_FWD_BF16, 125, C0, Co)
    prev = g16(rb(rng.standard_normal((B, D, H, W, Co))), dev)
    exact = O.conv_nd_fwd(x0, rb(w), 1) + prev.float().double().cpu().numpy()
    out = torch.empty_like(prev)
    ops._conv5_b16_call(tx, None, wp, None, out, None, (D, H, W), acc_src=prev)
    check_bf16("accumulate out of place %s" % (shape,), out, exact)
    inpl = prev.clone()
    ops._conv5_b16_call(tx, None, wp, None, inpl, None, (D, H, W), accum=True)
    assert torch.equal(inpl, out)


def test_deep_plan_is_what_the_docs_say(dev):
    """The shapes DESIGN / profiles/r04_layer_table.txt quote really take the deep kernel: statistics rows = 4x8x8 bricks without a K
    split, reduce blocks with one."""
    from vnet_tensorflow_amd import _lib
    L = _lib.lib()
    assert L.vnet_conv_b16_stats_rows(64, 0, 64, 0, 1, 32, 32, 32) == 8 * 4 * 4
    assert L.vnet_conv_b16_stats_rows(128, 0, 64, 0, 1, 32, 32, 32) == 8 * 4 * 4
    assert L.vnet_conv_b16_stats_rows(128, 0, 128, 0, 1, 16, 16, 16) == min(2048, 16 ** 3 * 128 // 256)
    assert L.vnet_conv_b16_stats_rows(256, 0, 256, 0, 1, 8, 8, 8) == min(2048, 8 ** 3 * 256 // 256)
    # 128^3 / 64^3 layers keep their kernels (row-pair / 16-cout): rows per their bricks
    assert L.vnet_conv_b16_stats_rows(32, 0, 32, 0, 1, 64, 64, 64) == 16 * 4 * 4
    assert L.vnet_conv_b16_ws_bytes(256, 0, 256, 0, 1, 8, 8, 8) >= 16 * 512 * 256 * 4

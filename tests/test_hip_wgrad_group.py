"""-m gpu: the grouped launch of the deep-level 5^3 filter gradients (include/vnet_hip.h: vnet_conv_wgrad_b16_group; reference: the
per-convolution gradient ops of model.py:660 for layers2.py:59-63, which TF schedules as independent nodes).

Bar: every layer of a group equals the fp64 oracle's filter gradient on the same bf16-valued tensors to fp32-accumulation accuracy
(the per-kernel bar of tests/test_hip_deep.py: 2e-6 of the largest entry) whatever the other layers of the group are, two launches
of the same group are bit-identical, and the whole training step with the grouped launch computes the gradients of the step without
it (a residual block's ds is the dy of a waiting filter gradient AND the tensor the block's first convolution used to add its
gradient into in place: the step must not change it before the group has run)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import vnet_oracle as O
from tests.test_hip_b16 import rb, g16, _conv5_inputs
from tests.util import g, check_close

pytestmark = pytest.mark.gpu

#          B, D,  H,  W,  C0, C1, Cout              kernel family inside the group
GROUP = [
    (1, 8, 16, 32, 32, 0, 32),      # row-reuse bricks (4 x 8 x 32): W >= 32
    (1, 4, 8, 32, 16, 16, 48),      # ... two sources, three 16-cout blocks
    (1, 16, 16, 16, 32, 0, 64),     # 4 x 4 x 16 bricks, two cout blocks per workgroup, two tap groups
    (2, 5, 9, 16, 16, 0, 32),       # ... ragged, batch 2
    (1, 8, 8, 8, 64, 0, 64),        # 4 x 8 x 8 bricks
    (1, 6, 7, 5, 32, 32, 32),       # ... ragged, two sources
    (1, 8, 8, 8, 16, 0, 16),        # 16 output channels at 8^3: not a shape of the grouped kernels -> the layer's own launch
]


def test_group_with_2cube_layers(dev):
    """The other members of a V-Net pass: the 2^3 stride-2 filter gradients (ks = 2 jobs: x = the fine tensor, dy = the coarse one;
    their dw is the filter gradient of the down conv and, with the roles of the tensors swapped by the caller, of the transposed conv)
    next to 5^3 layers.  (The zero-padded network input keeps its own launch: tests/test_hip_b16.py.)"""
    from vnet_tensorflow_amd import ops
    rng = np.random.default_rng(77)
    five = [(1, 8, 16, 32, 32, 0, 32), (1, 8, 8, 8, 64, 0, 64)]
    two = [(1, 16, 16, 32, 16, 32), (2, 9, 10, 12, 32, 64), (1, 8, 8, 8, 64, 128), (1, 16, 16, 16, 8, 16)]     # B, D, H, W (fine), Cin, Cout
    jobs = []
    with ops.deferred_wgrad_reduce():
        for k, shape in enumerate(five):
            B, D, H, W, C0, C1, Co = shape
            x0, x1, w, b, dy = _conv5_inputs(shape, 300 + k)
            dw = torch.full((5, 5, 5, C0, Co), float("nan"), dtype=torch.float32, device=dev)
            sink = ops.GradSink(dw)
            ops._wgrad5_b16_call(g16(x0, dev), None, g16(dy, dev), dw, (D, H, W), C0, owner=sink)
            jobs.append(("5^3 %s" % (shape,), dw, sink, O.conv_nd_bwd(x0, np.zeros((5, 5, 5, C0, Co)), dy, 1, need_dx=False)[1]))
        for (B, D, H, W, Ci, Co) in two:
            xf = rb(rng.standard_normal((B, D, H, W, Ci)))
            dc = tuple((d + 1) // 2 for d in (D, H, W))
            dyc = rb(rng.standard_normal((B,) + dc + (Co,)))
            dw = torch.full((2, 2, 2, Ci, Co), float("nan"), dtype=torch.float32, device=dev)
            sink = ops.GradSink(dw)
            ops._wgrad2_b16_call(g16(xf, dev), g16(dyc, dev), dw, (D, H, W), dc, owner=sink)
            jobs.append(("2^3 %s" % ((B, D, H, W, Ci, Co),), dw, sink, O.conv_nd_bwd(xf, np.zeros((2, 2, 2, Ci, Co)), dyc, 2, need_dx=False)[1]))
        assert len(ops._DEFER["jobs"]) == len(jobs)
    torch.cuda.synchronize()
    for name, dw, _, ex in jobs:
        check_close("group member " + name, dw, ex, 2e-6)


def _run_group(dev, shapes, rounds=None, lib_option=None):
    from vnet_tensorflow_amd import ops
    if rounds is not None:
        lib_option("WGRAD_GROUP_ROUNDS", rounds)
    ins, outs, sinks = [], [], []
    for k, shape in enumerate(shapes):
        B, D, H, W, C0, C1, Co = shape
        x0, x1, w, b, dy = _conv5_inputs(shape, 100 + 7 * k + sum(shape))
        ins.append((x0, x1, dy))
        dw = torch.full((5, 5, 5, C0 + C1, Co), float("nan"), dtype=torch.float32, device=dev)
        outs.append(dw)
        sinks.append(ops.GradSink(dw))
    with ops.deferred_wgrad_reduce():
        for shape, (x0, x1, dy), dw, sink in zip(shapes, ins, outs, sinks):
            B, D, H, W, C0, C1, Co = shape
            ops._wgrad5_b16_call(g16(x0, dev), g16(x1, dev) if C1 else None, g16(dy, dev), dw, (D, H, W), C0 + C1, owner=sink)
        assert len(ops._DEFER["jobs"]) == len(shapes)          # nothing has been launched yet
    assert not ops._DEFER["jobs"] and not ops._DEFER["dy_ptrs"]
    torch.cuda.synchronize()
    return ins, outs


@pytest.mark.parametrize("rounds", [None, "8", "0.25", "0"])
def test_group_members_against_oracle(dev, rounds, lib_option):
    """rounds: the plan's workgroups per CU -- 8 splits every layer over many workgroups (slabs + reduce everywhere), 0.25 leaves
    most layers unsplit (direct writes of dw), 0 launches every layer on its own."""
    ins, outs = _run_group(dev, GROUP, rounds, lib_option)
    for shape, (x0, x1, dy), dw in zip(GROUP, ins, outs):
        xcat = x0 if x1 is None else np.concatenate((x0, x1), -1)
        _, dw_ex = O.conv_nd_bwd(xcat, np.zeros((5, 5, 5, xcat.shape[-1], dy.shape[-1])), dy, 1)
        check_close("group rounds=%s %s dw" % (rounds, shape), dw, dw_ex, 2e-6)


def test_group_is_deterministic_and_independent_of_company(dev, monkeypatch):
    _, a = _run_group(dev, GROUP)
    _, b = _run_group(dev, GROUP)
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    # a layer's bits depend on how the plan splits IT, not on the other layers' data: the same plan with other company data
    # (here: the same shapes in another order changes nothing either -- the plan is a function of the set of shapes)
    order = [3, 0, 6, 5, 1, 4, 2]
    ins, c = _run_group(dev, [GROUP[k] for k in order])
    for k, q in zip(order, c):
        # (inputs are seeded by position, so compare against the oracle again rather than against `a`)
        x0, x1, dy = ins[order.index(k)]
        xcat = x0 if x1 is None else np.concatenate((x0, x1), -1)
        _, dw_ex = O.conv_nd_bwd(xcat, np.zeros((5, 5, 5, xcat.shape[-1], dy.shape[-1])), dy, 1)
        check_close("reordered group %s" % (GROUP[k],), q, dw_ex, 2e-6)


ZS_SHAPES = [
    (1, 8, 16, 32, 32, 0, 32),      # 32-wide columns, ring of eight planes wraps once
    (1, 20, 9, 37, 16, 16, 64),     # ... ragged, two sources, two cout blocks, 20 z steps: the ring wraps five times
    (1, 16, 16, 16, 32, 0, 64),     # 16-wide columns: two planes per step
    (2, 7, 9, 17, 16, 0, 32),       # ... ragged (odd depth: a half-empty plane pair), batch 2
    (1, 8, 8, 8, 64, 0, 64),        # 8-wide columns: four planes per step, the whole column resident
    (1, 6, 7, 5, 32, 32, 96),       # ... ragged, two sources, three cout blocks
    (1, 12, 4, 4, 16, 0, 32),       # the deepest column the 8-wide form takes
]


@pytest.mark.parametrize("shape", ZS_SHAPES)
def test_z_streaming_kernel_against_oracle(dev, shape, monkeypatch, lib_option):
    """csrc/wgrad_zs.h on its own (VNET_WGRAD_ZS=1 routes the per-layer entry point to it): column steps split over workgroups,
    slabs + reduce; and with a workspace of ONE slab (no split: every step of a (chunk, cout block) in one workgroup)."""
    from vnet_tensorflow_amd import ops, _lib
    lib_option("WGRAD_ZS", "1")
    B, D, H, W, C0, C1, Co = shape
    x0, x1, w, b, dy = _conv5_inputs(shape, sum(shape) + 17)
    xcat = x0 if x1 is None else np.concatenate((x0, x1), -1)
    _, dw_ex = O.conv_nd_bwd(xcat, np.zeros((5, 5, 5, C0 + C1, Co)), dy, 1)
    tx0, tx1, tdy = g16(x0, dev), (g16(x1, dev) if C1 else None), g16(dy, dev)
    dw = torch.full((5, 5, 5, C0 + C1, Co), float("nan"), dtype=torch.float32, device=dev)
    ops._wgrad5_b16_call(tx0, tx1, tdy, dw, (D, H, W), C0 + C1)
    check_close("z-streaming %s dw" % (shape,), dw, dw_ex, 2e-6)
    dw2 = torch.full_like(dw, float("nan"))
    ops._wgrad5_b16_call(tx0, tx1, tdy, dw2, (D, H, W), C0 + C1)
    assert torch.equal(dw, dw2)
    # one slab of workspace: nsplit = 1
    L = _lib.lib()
    slab = 125 * (-(-(C0 + C1) // 16) * 16) * (-(-Co // 16) * 16) * 4
    ws = torch.empty(slab, dtype=torch.uint8, device=dev)
    dw3 = torch.full_like(dw, float("nan"))
    _lib.check(L.vnet_conv_wgrad_b16(ops._ptr(tx0), C0, ops._ptr(tx1), C1, ops._ptr(tdy), Co, ops._ptr(dw3), C0 + C1, B, D, H, W,
                                     ops._ptr(ws), slab, None), "vnet_conv_wgrad_b16")
    torch.cuda.synchronize()
    check_close("z-streaming unsplit %s dw" % (shape,), dw3, dw_ex, 2e-6)


def test_c_abi_rejects_bad_jobs(dev):
    from vnet_tensorflow_amd import _lib
    L = _lib.lib()
    assert L.vnet_conv_wgrad_b16_group(None, 0, None) == 0
    assert L.vnet_conv_wgrad_b16_group(None, 1, None) == -1
    arr = (_lib.WgradJob * 1)()
    assert L.vnet_conv_wgrad_b16_group(ctypes.addressof(arr), 1, None) == -1          # null tensors


@pytest.mark.parametrize("P", [16, 32])
def test_training_step_gradients_do_not_depend_on_the_grouping(dev, P, monkeypatch):
    """One fwd + bwd of a bf16-storage V-Net with two- and three-convolution residual blocks (their ds is the dy of the block's last
    filter gradient and the accumulation target of its first convolution's backward-data), grouped vs every layer on its own: the
    data gradients are bit-identical by construction (out-of-place accumulate = the same arithmetic), the filter gradients agree to
    summation order."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd.model import image2label
    from oracle.vnet_oracle import synthetic_batch
    from tests.test_hip_train_loop import _cfg
    import pathlib
    monkeypatch.setenv("VNET_STEP_GRAPH", "0")
    x, lab = synthetic_batch(1, P, 1, 2, seed=5)
    xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
    res, grouped = {}, []
    flush = ops._flush_wgrad_group
    monkeypatch.setattr(ops, "_flush_wgrad_group", lambda launch=True: (grouped.append(len(ops._DEFER["jobs"])), flush(launch))[1])
    for on in (True, False):
        ops.set_wgrad_group(on)
        try:
            np.random.seed(9)
            cfg = _cfg(pathlib.Path("/tmp"), PatchShape=[P] * 3, BatchSize=1)
            cfg["TrainingSetting"]["Networks"].update(NumChannel=16, NumLevels=3, NumConvolutions=[1, 2, 3], BottomConvolutions=2)
            cfg["TrainingSetting"]["ComputeDtype"] = "bf16"
            m = image2label(None, cfg, device=dev, verbose=False)
            m.read_config(); m.build_model_graph(); m._setup_training()
            with ops.context(m.ctx):
                loss = m._compute_gradients(xt, lt, 0.0)
            torch.cuda.synchronize()
            res[on] = (float(loss), m.flat.grad.clone())
        finally:
            ops.set_wgrad_group(True)
    assert res[True][0] == res[False][0]
    ga, gb = res[True][1], res[False][1]
    assert torch.isfinite(ga).all()
    scale = float(gb.abs().max())
    assert float((ga - gb).abs().max()) <= 2e-5 * scale, (float((ga - gb).abs().max()), scale)
    assert max(grouped) >= 10 and min(grouped) == 0, grouped     # one pass collected the 5^3 layers, the other none
    # byte budget: with a small budget the pass launches its group in several pieces; the same gradients to summation order
    monkeypatch.setitem(ops._GROUP, "max_bytes", 3 * P ** 3 * 16 * 2)
    grouped.clear()
    np.random.seed(9)
    m = image2label(None, cfg, device=dev, verbose=False)
    m.read_config(); m.build_model_graph(); m._setup_training()
    with ops.context(m.ctx):
        loss = m._compute_gradients(xt, lt, 0.0)
    torch.cuda.synchronize()
    assert float(loss) == res[False][0]
    assert float((m.flat.grad - gb).abs().max()) <= 2e-5 * scale
    assert len([g for g in grouped if g > 0]) >= 3, grouped         # several partial groups instead of one



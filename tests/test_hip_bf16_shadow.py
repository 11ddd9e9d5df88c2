"""-m gpu: bf16 shadows of the activations (include/vnet_hip.h *_x16; ops._alloc_shadowed).  In bf16 compute mode the
batch-norm / dropout kernels write a bf16 image of their output behind the fp32 tensor and the 5^3 convolutions stage that
image.  The image is the very rounding the convolutions apply themselves (RNE), so every result must be BIT-identical with
the shadows switched off -- which is what ties this path to the oracle-checked one (test_hip_ops.py::test_conv5_bf16,
the bf16 golden networks)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _shadowed(ops, t):
    y = ops.with_shadow(t)
    assert ops._shadow_ptr(y) == y.data_ptr() + 4 * t.numel()
    return y


@pytest.mark.parametrize("shape,cin,cin1,cout", [
    ((1, 32, 32, 32), 16, 0, 16),      # 16-cout persistent kernel
    ((1, 16, 40, 48), 16, 16, 16),     # two sources, two chunks (paired schedule), ragged bricks
    ((1, 32, 32, 32), 16, 0, 32),      # 32-cout kernel, wide bricks
    ((2, 12, 20, 24), 32, 32, 64),     # half bricks, NSB 2
    ((1, 8, 8, 8), 64, 0, 128),        # small bricks, split-K
    ((1, 7, 9, 5), 8, 8, 24),          # 8-channel sources, ragged everything
    ((1, 64, 64, 64), 16, 0, 32),      # row-pair kernel (32-cout blocks, >= 256 items): one chunk
    ((1, 36, 70, 50), 8, 0, 64),       # row-pair kernel: ragged bricks, half-filled chunk, two cout blocks
    ((1, 32, 64, 64), 16, 16, 32),     # row-pair kernel: two sources = two chunks per brick
])
def test_conv_and_wgrad_from_shadows_bit_identical(dev, shape, cin, cin1, cout):
    from vnet_tensorflow_amd import ops
    ops.set_compute_dtype("bf16_operands")
    try:
        gen = torch.Generator(device="cpu").manual_seed(cin * 131 + cout)
        B, D, H, W = shape
        x0 = torch.randn(B, D, H, W, cin, generator=gen).to(dev)
        x1 = torch.randn(B, D, H, W, cin1, generator=gen).to(dev) if cin1 else None
        dy = torch.randn(B, D, H, W, cout, generator=gen).to(dev)
        w = (torch.randn(5, 5, 5, cin + cin1, cout, generator=gen) * 0.05).to(dev)
        b = torch.randn(cout, generator=gen).to(dev)
        wp = ops.packed_weights(w, ops.PACK_FWD_BF16, 125, cin + cin1, cout)
        ref = torch.empty(B, D, H, W, cout, device=dev)
        ops._conv_bf16_call(x0, x1, wp, b, ref, None, (D, H, W))
        got = torch.empty_like(ref)
        s0, s1, sdy = _shadowed(ops, x0), (_shadowed(ops, x1) if cin1 else None), _shadowed(ops, dy)
        ops._conv_bf16_call(s0, s1, wp, b, got, None, (D, H, W))
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        # accumulate mode
        acc_ref, acc_got = ref.clone(), ref.clone()
        ops._conv_bf16_call(x0, x1, wp, b, acc_ref, None, (D, H, W), accum=True)
        ops._conv_bf16_call(s0, s1, wp, b, acc_got, None, (D, H, W), accum=True)
        assert torch.equal(acc_got, acc_ref)
        # filter gradient
        dw_ref = torch.empty_like(w)
        dw_got = torch.empty_like(w)
        ops._wgrad_bf16_call(x0, x1, dy, dw_ref, (D, H, W))
        ops._wgrad_bf16_call(s0, s1, sdy, dw_got, (D, H, W))
        torch.cuda.synchronize()
        assert torch.equal(dw_got, dw_ref)
        # a shadow that differs from the fp32 data must show (the x16 path really reads the shadow)
        n = s0.numel()
        sh = torch.empty(0, dtype=torch.bfloat16, device=dev).set_(s0.untyped_storage(), 2 * n, (n,))
        sh.mul_(2.0)
        ops._conv_bf16_call(s0, s1, wp, None, got, None, (D, H, W))
        assert not torch.equal(got, ref)
    finally:
        ops.set_compute_dtype("fp32")


def test_producers_write_rne_shadows(dev):
    from vnet_tensorflow_amd import ops
    ops.set_compute_dtype("bf16_operands")
    try:
        gen = torch.Generator(device="cpu").manual_seed(5)
        x = torch.randn(1, 6, 10, 12, 16, generator=gen).to(dev).requires_grad_(True)
        r = torch.randn(1, 6, 10, 12, 16, generator=gen).to(dev)
        gamma = (1 + 0.2 * torch.randn(16, generator=gen)).to(dev).requires_grad_(True)
        beta = (0.2 * torch.randn(16, generator=gen)).to(dev).requires_grad_(True)
        alpha = (0.1 * torch.rand(16, generator=gen)).to(dev).requires_grad_(True)
        x._vnet_dy16 = True
        y = ops.bn_act(x, gamma, beta, "prelu", alpha, residual=r)
        n = y.numel()
        assert ops._shadow_ptr(y) is not None
        sh = torch.empty(0, dtype=torch.bfloat16, device=dev).set_(y.untyped_storage(), 2 * n, (n,))
        assert torch.equal(sh, y.detach().reshape(-1).to(torch.bfloat16))
        seen = {}
        x.register_hook(lambda gr: seen.setdefault("g", gr))
        y.backward(torch.randn(y.shape, generator=gen).to(dev))
        ds = seen["g"]
        assert ops._shadow_ptr(ds) is not None
        sh = torch.empty(0, dtype=torch.bfloat16, device=dev).set_(ds.untyped_storage(), 2 * n, (n,))
        assert torch.equal(sh, ds.reshape(-1).to(torch.bfloat16))
        # dropout
        d = ops.dropout(y.detach(), 0.25)
        assert ops._shadow_ptr(d) is not None
        sh = torch.empty(0, dtype=torch.bfloat16, device=dev).set_(d.untyped_storage(), 2 * n, (n,))
        assert torch.equal(sh, d.reshape(-1).to(torch.bfloat16))
        # a look-alike is not a shadow: first two of three slabs = offset 0, contiguous, storage exactly 1.5x the view
        look = torch.randn(3, 4, 4, 4, 16, device=dev)[:2]
        assert look.untyped_storage().nbytes() == look.numel() * 6 and ops._shadow_ptr(look) is None
        # fp32 mode: plain allocations
        ops.set_compute_dtype("fp32")
        y32 = ops.bn_act(x.detach(), gamma.detach(), beta.detach(), "prelu", alpha.detach())
        assert ops._shadow_ptr(y32) is None
    finally:
        ops.set_compute_dtype("fp32")


@pytest.mark.parametrize("variant,cin,dropout", [("networks", 4, 0.0), ("networks", 1, 0.0), ("VNet", 2, 0.0), ("networks", 3, 0.15)])
def test_network_step_bit_identical_with_and_without_shadows(dev, variant, cin, dropout):
    """Whole forward + loss + backward of a bf16-mode network: logits, loss and every gradient bit-identical."""
    from vnet_tensorflow_amd import ops, networks, VNet
    from oracle.vnet_oracle import synthetic_batch
    K, C0, levels, ncv, nb = 3, 8, 2, [1, 2], 2
    x, lab = synthetic_batch(2, 16, cin, K, seed=11)
    results = []
    ops.set_compute_dtype("bf16_operands")
    try:
        for on in (True, False):
            ops.set_bf16_shadows(on)
            torch.manual_seed(0)
            np.random.seed(0)
            ops._DROP_SEED[0] = 0x5EED            # the same dropout masks in both runs (the dropout kernel writes shadows, too)
            if variant == "networks":
                net = networks.VNet(K, dropout, C0, levels, ncv, nb, True, "prelu", device=dev)
            else:
                net = VNet.VNet(K, 1.0 - dropout, C0, levels, ncv, nb, True, "prelu", device=dev)
            net.build(x.shape)
            with torch.no_grad():
                gen = torch.Generator().manual_seed(1)
                for name, p in net.named_parameters():
                    if not name.endswith("weights"):
                        p.add_(0.2 * torch.randn(p.shape, generator=gen).to(dev))
            xs = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
            logits = net.GetNetwork(xs) if variant == "networks" else net.network_fn(xs)
            l, dice = ops.softmax_loss(logits, torch.from_numpy(lab.astype(np.int32)).to(dev), "sorensen", [1.0] * K, 0.7)[:2]
            l.backward()
            torch.cuda.synchronize()
            results.append((logits.detach().clone(), float(l.detach()),
                            {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}))
    finally:
        ops.set_bf16_shadows(True)
        ops.set_compute_dtype("fp32")
    (la, lossa, ga), (lb, lossb, gb) = results
    assert torch.equal(la, lb)
    assert lossa == lossb
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n
        if ga[n] is not None:
            assert torch.equal(ga[n], gb[n]), n


@pytest.mark.parametrize("shape,cin,cout,residual", [((1, 64, 64, 64), 32, 32, True), ((1, 36, 70, 50), 16, 64, False)])
def test_row_pair_kernel_epilogue_statistics(dev, shape, cin, cout, residual):
    """conv5_bf16_r32_kernel<STATS>: one partial row [sum | sum of squares] per 4x16x16 brick (vnet_conv_bf16_stats_rows_x16),
    of y (+ residual), voxels outside a ragged volume not counted; the output itself is the plain launch's, bit for bit."""
    from vnet_tensorflow_amd import ops
    from vnet_tensorflow_amd._lib import lib
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(cin + cout)
    x = ops.with_shadow((torch.randn(B, D, H, W, cin, generator=gen) * 1.5 + 0.2).to(dev))
    w = (torch.randn(5, 5, 5, cin, cout, generator=gen) * 0.05).to(dev)
    b = torch.randn(cout, generator=gen).to(dev)
    res = (torch.randn(B, D, H, W, cout, generator=gen) * 2.0).to(dev) if residual else None
    ops.set_compute_dtype("bf16_operands")
    try:
        rows = lib().vnet_conv_bf16_stats_rows_x16(cin, cout, 0, cin, 0, B, D, H, W)
        assert rows == B * -(-D // 4) * -(-H // 16) * -(-W // 16)
        with torch.no_grad():
            y = ops.conv(x, w, b, 5, bn_stats=True, bn_residual=res)
            st = y._vnet_stats
            assert st.rows == rows and tuple(st.partial.shape) == (rows, 2 * cout)
            plain = ops.conv(x, w, b, 5)
        assert torch.equal(y, plain)
        v = (y + res if residual else y).double().reshape(-1, cout)
        tot = st.partial.double().sum(0)
        assert torch.allclose(tot[:cout], v.sum(0), rtol=1e-5, atol=1e-2)
        assert torch.allclose(tot[cout:], (v * v).sum(0), rtol=1e-5, atol=1e-2)
    finally:
        ops.set_compute_dtype("fp32")


def test_row_pair_kernel_random_shapes_bit_identical_to_generic(dev):
    """Property test: on random ragged volumes with >= 256 (brick, cout block) items the shadow path takes the row-pair kernel
    and must reproduce the generic fp32-source kernel bit for bit (same products, same order), forward and accumulate mode."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from vnet_tensorflow_amd import ops

    @settings(max_examples=12, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(D=st.integers(33, 72), H=st.integers(49, 80), W=st.integers(49, 80), cin=st.sampled_from([8, 16, 24, 32]),
           cout=st.sampled_from([32, 64]), two=st.booleans(), seed=st.integers(0, 10 ** 6))
    def run(D, H, W, cin, cout, two, seed):
        gen = torch.Generator().manual_seed(seed)
        x0 = torch.randn(1, D, H, W, cin, generator=gen).to(dev)
        x1 = torch.randn(1, D, H, W, 8, generator=gen).to(dev) if two else None
        w = (torch.randn(5, 5, 5, cin + (8 if two else 0), cout, generator=gen) * 0.05).to(dev)
        b = torch.randn(cout, generator=gen).to(dev)
        wp = ops.packed_weights(w, ops.PACK_FWD_BF16, 125, cin + (8 if two else 0), cout)
        ref = torch.randn(1, D, H, W, cout, generator=gen).to(dev)
        got = ref.clone()
        ops._conv_bf16_call(x0, x1, wp, b, ref, None, (D, H, W), accum=True)
        ops._conv_bf16_call(ops.with_shadow(x0), ops.with_shadow(x1) if two else None, wp, b, got, None, (D, H, W), accum=True)
        assert torch.equal(got, ref), (D, H, W, cin, cout, two)

    ops.set_compute_dtype("bf16_operands")
    try:
        run()
    finally:
        ops.set_compute_dtype("fp32")

"""Streaming batch-norm passes, one kernel at a time, at the tensor shapes of the 128^3 network (16 channels at 128^3 ... 256 at
8^3), both storage modes:   python profiles/bench_bn.py [iters]
Each figure = `iters` back-to-back launches of ONE kernel between two HIP events on the launch stream (C ABI called directly, as
ops.py does), GB/s over the ALGORITHMIC bytes of the pass (what it must read and write once):
    statistics      read x                          (only where the producing conv cannot write them in its epilogue)
    normalise+act   read x (+ residual), write y
    bwd reduce      read dy, x (+ residual)
    bwd apply       read dy, x (+ residual), write ds
"""
import sys

import torch

sys.path.insert(0, '.')
from vnet_tensorflow_amd import _lib, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda', 0)
L = _lib.lib()
st = ops._stream()
P = ops._ptr


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


print("%-6s %-18s %9s %9s %9s %9s   (us per launch | GB/s algorithmic), PReLU, %d back-to-back launches" % ("mode", "tensor", "stats", "fwd", "reduce", "apply", iters))
for mode in ("fp32", "bf16"):
    b16 = mode == "bf16"
    dt = torch.bfloat16 if b16 else torch.float32
    esz = 2 if b16 else 4
    for res in (False, True):
        for n, C in ((128, 16), (64, 32), (32, 64), (16, 128), (8, 256)):
            M = n ** 3
            x = torch.randn(M, C, device=dev).to(dt)
            r = torch.randn(M, C, device=dev).to(dt) if res else None
            dy = torch.randn(M, C, device=dev).to(dt)
            y = torch.empty_like(x)
            ds = torch.empty_like(x)
            mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev)
            gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev); alpha = torch.full((C,), 0.25, device=dev)
            dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); da = torch.zeros(C, device=dev)
            mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
            nb = L.vnet_bn_ws_bytes(C)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            if b16:
                f_stats = lambda: L.vnet_bn_stats_b16(P(x), P(r), M, C, 1e-3, 0.99, P(mean), P(invstd), P(mm), P(mv), P(ws), nb, st)
                f_fwd = lambda: L.vnet_bn_act_fwd_b16(P(x), P(r), 0, M, C, P(mean), P(invstd), P(gamma), P(beta), 2, P(alpha), P(y), st)
                f_red = lambda: L.vnet_bn_act_bwd_reduce_b16(P(dy), P(x), P(r), 0, M, C, P(mean), P(invstd), P(gamma), P(beta), 2, P(alpha),
                                                             P(dg), P(db), P(da), P(ws), nb, st)
                f_app = lambda: L.vnet_bn_act_bwd_apply_b16(P(dy), P(x), P(r), 0, M, C, P(mean), P(invstd), P(gamma), P(beta), 2, P(alpha),
                                                            P(db), P(dg), float(M), None, P(ds), st)
            else:
                f_stats = lambda: L.vnet_bn_stats(P(x), P(r), 0, M, C, 1e-3, 0.99, P(mean), P(invstd), P(mm), P(mv), P(ws), nb, st)
                f_fwd = lambda: L.vnet_bn_act_fwd(P(x), P(r), 0, M, C, P(mean), P(invstd), P(gamma), P(beta), 2, P(alpha), P(y), st)
                f_red = lambda: L.vnet_bn_act_bwd_reduce(P(dy), P(x), P(r), 0, M, C, P(mean), P(invstd), P(gamma), P(beta), 2, P(alpha),
                                                         P(dg), P(db), P(da), P(ws), nb, st)
                f_app = lambda: L.vnet_bn_act_bwd_apply(P(dy), P(x), P(r), 0, M, C, P(mean), P(invstd), P(gamma), P(beta), 2, P(alpha),
                                                        P(db), P(dg), float(M), None, P(ds), st)
            nr = 2 if res else 1
            byts = {"stats": nr * M * C * esz, "fwd": (nr + 1) * M * C * esz, "reduce": (nr + 1) * M * C * esz, "apply": (nr + 2) * M * C * esz}
            out = []
            for tag, fn in (("stats", f_stats), ("fwd", f_fwd), ("reduce", f_red), ("apply", f_app)):
                us = timed(fn)
                out.append("%6.1f|%5.0f" % (us, byts[tag] / us / 1e3))
            print("%-6s %-18s %s" % (mode, "%d^3 x %d%s" % (n, C, " +res" if res else ""), "  ".join(out)))

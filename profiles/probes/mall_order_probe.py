"""Does the traversal order of a pass matter to the pass behind it (256 MiB Infinity Cache)?  Pass A reads two tensors front to back
(torch.dot); pass B (out = a * b, in 16 slabs along the leading axis) runs its slabs in the same order or in reverse.  Sizes: the 128^3
tensors of the V-Net step (16 channels: 134 MB each; 32 channels: 268 MB each).
python profiles/probes/mall_order_probe.py"""
import torch
dev = torch.device("cuda", 0)
for C in (8, 16, 32):
    n = 128 ** 3 * C
    a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); out = torch.empty(n, device=dev)
    S = 16
    sl = [slice(i * (n // S), (i + 1) * (n // S)) for i in range(S)]
    def run(order, reps=20):
        tot = 0.0
        for _ in range(reps):
            torch.dot(a, b)                                   # pass A, front to back
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in order:
                torch.mul(a[sl[i]], b[sl[i]], out=out[sl[i]])
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e3
    run(range(S), 3)
    f1, r1, f2, r2 = run(range(S)), run(range(S - 1, -1, -1)), run(range(S)), run(range(S - 1, -1, -1))
    print("128^3 x %2d ch (%4d MB per tensor): pass B after pass A  same order %.1f / %.1f us   reversed %.1f / %.1f us" % (C, n * 4 >> 20, f1, f2, r1, r2))

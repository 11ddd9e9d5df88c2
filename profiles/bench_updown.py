"""2^3 stride-2 down convolution and 2^3 transposed convolution alone, at the four level transitions of the 128^3 net:
   python profiles/bench_updown.py      (GB/s = algorithmic bytes: input + output tensors)"""
import sys, torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops
dev = torch.device('cuda', 0)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for P, c in ((128, 16), (64, 32), (32, 64), (16, 128)):
    x = torch.randn(1, P, P, P, c, device=dev)
    wd = torch.randn(2, 2, 2, c, 2 * c, device=dev) * 0.05
    bd = torch.zeros(2 * c, device=dev)
    xu = torch.randn(1, P // 2, P // 2, P // 2, 2 * c, device=dev)
    wu = torch.randn(2, 2, 2, c, 2 * c, device=dev) * 0.05
    bu = torch.zeros(c, device=dev)
    nbytes = 4.0 * (x.numel() + xu.numel())
    with torch.no_grad():
        td = timed(lambda: ops.conv(x, wd, bd, 2, stride=2))
        tu = timed(lambda: ops.conv_transpose2(xu, wu, bu, (P, P, P)))
    print("%3d^3 x%-3d  down %6.1f us %5.2f TB/s   up %6.1f us %5.2f TB/s" % (P, c, td, nbytes / td / 1e6, tu, nbytes / tu / 1e6))

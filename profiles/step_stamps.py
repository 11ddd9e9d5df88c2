"""s_memtime stamps of four consecutive brick steps of one workgroup of a persistent bf16-storage conv kernel.
   bash profiles/build_stamps.sh          (here: -DVNET_STAMPS experiment build -> profiles/probes/libvnet_hip_stamps.so)
   VNET_HIP_LIB=$PWD/profiles/probes/libvnet_hip_stamps.so python profiles/step_stamps.py <c16|r32> Cin Cout [stats]
Prints cycles per phase per wave; waves w and w+4 share a SIMD.  The shipped library has no stamps (the script refuses it)."""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import _lib, ops

kind, ci, co = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
want_stats = len(sys.argv) > 4 and sys.argv[4] == "stats"
dev = torch.device('cuda', 0)
ops.set_compute_dtype('bf16')
L = _lib.lib()
if not hasattr(L, "vnet_debug_set_stamps"):
    sys.exit("needs the -DVNET_STAMPS build in VNET_HIP_LIB")
gen = torch.Generator().manual_seed(1)
P = 128
shp = (1, P, P, P)
if ci == 32:
    x0 = torch.randn(*shp, 16, generator=gen).to(dev).to(torch.bfloat16)
    x1 = torch.randn(*shp, 16, generator=gen).to(dev).to(torch.bfloat16)
else:
    x0, x1 = torch.randn(*shp, ci, generator=gen).to(dev).to(torch.bfloat16), None
w = (torch.randn(5, 5, 5, ci, co, generator=gen) * 0.05).to(dev)
b = torch.randn(co, generator=gen).to(dev)
wp = ops.packed_weights(w, ops.PACK_FWD_BF16, 125, ci, co)
y = torch.empty(*shp, co, device=dev, dtype=torch.bfloat16)
stats = None
if want_stats:
    rows = L.vnet_conv_b16_stats_rows(min(ci, 16), ci - min(ci, 16), co, 0, 1, P, P, P)
    stats = torch.zeros(rows, 2, co, device=dev)
dbg = torch.zeros(4 * 8 * 12, dtype=torch.int64, device=dev)
fn = L.vnet_debug_set_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(dbg.data_ptr()) == 0
for _ in range(3):
    ops._conv5_b16_call(x0, x1, wp, b, y, None, shp[1:], stats=stats)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops._conv5_b16_call(x0, x1, wp, b, y, None, shp[1:], stats=stats)
e1.record()
torch.cuda.synchronize()
print("%s %d->%d%s  %.3f ms per launch" % (kind, ci, co, " +stats" if want_stats else "", e0.elapsed_time(e1) / 10))
t = dbg.cpu().numpy().reshape(4, 8, 12)
if kind == "c16":
    names = ["tile_issue", "dz pairs (200 MFMA)", "pack + dz=4 (60 MFMA)", "epilogue", "barrier", "commit(+filter)", "barrier"]
else:
    names = ["tile_issue", "plane0", "plane1", "plane2", "plane3", "plane4", "epilogue", "barrier", "commit", "barrier"]
n = len(names)
for st in range(4):
    print("step", 4 + st)
    for wv in range(8):
        r = t[st, wv]
        print("  wave", wv, "  ".join("%s %5d" % (nm, int(r[k + 1] - r[k])) for k, nm in enumerate(names)), "  total", int(r[n] - r[0]))

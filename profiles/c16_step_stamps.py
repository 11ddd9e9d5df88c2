"""Prints the s_memtime stamps of one brick step of conv5_bf16_c16_kernel (profiles/r02_c16_step_stamps.txt).  Needs the
experiment build: the stamp patch of DESIGN section 8 applied to csrc/conv_mfma.hip (ts[] = __builtin_readcyclecounter() around
the phases, written through ConvArgs.stats for one workgroup), compiled with -DVNET_PLAN_ENV into a library named by VNET_HIP_LIB,
and VNET_C16_DBG=1 so that the plain kernel runs with the debug buffer in `stats`.  With the shipped library it prints zeros."""
import sys, torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops
dev = torch.device('cuda', 0)
ops.set_compute_dtype('bf16')
gen = torch.Generator().manual_seed(1)
shp, ci, co = (1, 128, 128, 128), 16, 16
x = ops.with_shadow(torch.randn(*shp, ci, generator=gen).to(dev))
w = (torch.randn(5, 5, 5, ci, co, generator=gen) * 0.05).to(dev)
b = torch.randn(co, generator=gen).to(dev)
wp = ops.packed_weights(w, ops.PACK_FWD_BF16, 125, ci, co)
y = torch.empty(*shp, co, device=dev)
dbg = torch.zeros(4096 * 32, dtype=torch.float32, device=dev)
for _ in range(3):
    ops._conv_bf16_call(x, None, wp, b, y, None, shp[1:], stats=dbg)
torch.cuda.synchronize()
t = dbg.view(torch.int64)[:64].cpu().numpy().reshape(8, 8)
names = ["tile_issue", "dz pairs (200 MFMA)", "pack + dz=4 (60 MFMA)", "epilogue", "barrier", "commit", "barrier"]
for wv in range(8):
    r = t[wv]
    print("wave", wv, " ".join("%s %d" % (n, int(r[k + 1] - r[k])) for k, n in enumerate(names)), "total", int(r[7] - r[0]))

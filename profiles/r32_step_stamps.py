"""Prints the s_memtime stamps of one brick step of conv5_bf16_r32_kernel (profiles/r02_dvfs_probe.txt, last section).  Needs an
experiment build like profiles/c16_step_stamps.py: ts[] = __builtin_readcyclecounter() around the phases of the kernel, written
through ConvArgs.stats for one workgroup; -DVNET_PLAN_ENV library in VNET_HIP_LIB, VNET_C16_DBG=1.  Zeros with the shipped library."""
import sys, torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops
dev = torch.device('cuda', 0)
ops.set_compute_dtype('bf16')
gen = torch.Generator().manual_seed(1)
shp, ci, co = (1, 128, 128, 128), 16, 32
x = ops.with_shadow(torch.randn(*shp, ci, generator=gen).to(dev))
w = (torch.randn(5, 5, 5, ci, co, generator=gen) * 0.05).to(dev)
wp = ops.packed_weights(w, ops.PACK_FWD_BF16, 125, ci, co)
y = torch.empty(*shp, co, device=dev)
dbg = torch.zeros(4096 * 64, dtype=torch.float32, device=dev)
for _ in range(3):
    ops._conv_bf16_call(x, None, wp, None, y, None, shp[1:], stats=dbg)
torch.cuda.synchronize()
t = dbg.view(torch.int64)[:96].cpu().numpy().reshape(8, 12)
names = ["tile_issue", "plane0", "plane1", "plane2", "plane3", "plane4", "epilogue", "barrier", "commit", "barrier"]
for wv in range(8):
    r = t[wv]
    print("wave", wv, " ".join("%s %d" % (n, int(r[k + 1] - r[k])) for k, n in enumerate(names)), "total", int(r[10] - r[0]))

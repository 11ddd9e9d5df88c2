"""Prints the s_memtime stamps of conv5_x3_kernel (experiment build, profiles/x3_stamps.sh): cycles per phase of four consecutive
(item, chunk) steps of workgroup 88, per wave.   python profiles/x3_stamps.py [P Cin Cout]"""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from vnet_tensorflow_amd import ops, _lib
P, ci, co = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 32, 16)
dev = torch.device('cuda', 0)
ops.set_compute_dtype('fp32_split3')
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(4 * 8 * 12, dtype=torch.int64, device=dev)
raw.vnet_debug_set_stamps_x3.argtypes = [ctypes.c_void_p]
assert raw.vnet_debug_set_stamps_x3(buf.data_ptr()) == 0
x = torch.randn(1, P, P, P, ci, device=dev)
w = torch.randn(5, 5, 5, ci, co, device=dev) * 0.05
y = torch.empty(1, P, P, P, co, device=dev)
wp = ops.packed_weights(w, ops.PACK_FWD_X3, 125, ci, co)
for _ in range(3):
    ops._conv_x3_call(x, None, wp, None, y, None, (P, P, P))
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(4, 8, 12)
names = ["A+issue+P1", "P2", "P3(y)", "P4(col 4)", "barrier", "reduce+epilogue", "commit", "barrier"]
for st in range(4):
    for wv in range(8):
        r = t[st, wv]
        print("step", st + 4, "wave", wv, " ".join("%s %d" % (n, int(r[k + 1] - r[k])) for k, n in enumerate(names)), "| total", int(r[8] - r[0]),
              "| start skew", int(r[0] - t[st, 0, 0]))
    print()

"""Probe: can a HIP event recorded inside a stream capture be timed from outside the graph after a replay?
(hipEventRecordWithFlags(external) returns hipErrorInvalidValue under capture with torch 2.10's bundled ROCm 7.0 runtime.)"""
import ctypes
import torch

h = ctypes.CDLL("libamdhip64.so")
h.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
h.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
h.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
h.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
h.hipEventSynchronize.argtypes = [ctypes.c_void_p]
x = torch.randn(4096, 4096, device="cuda")
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
print("create", h.hipEventCreate(ctypes.byref(e0)), h.hipEventCreate(ctypes.byref(e1)))
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    raw = torch._C._cuda_getCurrentRawStream(0)
    r0 = h.hipEventRecordWithFlags(e0, raw, 1)
    if r0 != 0:
        h.hipGetLastError()                      # clear the sticky error of the refused call
        r0 = (r0, h.hipEventRecord(e0, raw))
    y = x * 2.0 + 1.0
    r1 = h.hipEventRecord(e1, raw)
print("record codes", r0, r1)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    print("elapsed rc", h.hipEventElapsedTime(ctypes.byref(ms), e0, e1), ms.value)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); y = x * 2.0 + 1.0; b.record(); torch.cuda.synchronize(); print("eager ms", a.elapsed_time(b))

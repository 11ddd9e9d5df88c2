// Test helper (not part of the product library): occupies `blocks` compute units with spinning workgroups for roughly
// `ms` milliseconds on the given stream -- a stand-in for a concurrent RCCL collective when judging how sensitive the
// kernels' grid sizes are to losing CUs.  Each workgroup claims 64 KB of LDS + 1024 threads so that it does not share
// a CU with anything substantial.
#include <hip/hip_runtime.h>
extern "C" {
__global__ void __launch_bounds__(1024) spin_kernel(long long cycles, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(32); }
    if (lds[(threadIdx.x * 7) & 1023] == -1) *sink = 1;
}
int cu_steal(int blocks, double ms, void* stream) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    static int* sink = nullptr;
    if (!sink) hipMalloc(&sink, 4);
    const long long cycles = (long long)(ms * 1e-3 * 100e6);          // wall_clock64 ticks at 100 MHz
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(1024), 100 * 1024, (hipStream_t)stream, cycles, sink);
    return (int)hipGetLastError();
}
}

// Test helper (not part of the product library): a stand-in for RCCL's gfx950 all-reduce kernel when judging, on ONE GPU, what a
// resident collective costs the backward kernels it overlaps.  Shape read from RCCL's code object: 256-thread
// workgroups, ~280 VGPRs, 19.7 KB LDS, one workgroup per channel.  It spins (s_sleep) for `ms` milliseconds: it holds its CU
// slots like the collective does but moves no data, so HBM / xGMI contention is NOT modelled -- an emulation, not a measurement.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o librccl_like.so rccl_like.hip
#include <hip/hip_runtime.h>
extern "C" {
__global__ void __launch_bounds__(256) rccl_like_kernel(long long ticks, int* sink) {
    __shared__ int lds[19712 / 4];
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a23, 0" ::: "v255", "a23");      // 256 VGPRs + 24 AGPRs = 280 registers, like the collective
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(16); }
    if (lds[(threadIdx.x * 7) & 255] == -1) *sink = 1;
}
int rccl_like(int channels, double ms, void* stream) {
    static int* sink = nullptr;
    if (!sink && hipMalloc(&sink, 4) != hipSuccess) return -1;
    const long long ticks = (long long)(ms * 1e-3 * 100e6);           // wall_clock64 ticks at 100 MHz
    hipLaunchKernelGGL(rccl_like_kernel, dim3(channels), dim3(256), 0, (hipStream_t)stream, ticks, sink);
    return (int)hipGetLastError();
}
}

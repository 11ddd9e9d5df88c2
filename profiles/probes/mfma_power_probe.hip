// Sustained rate of back-to-back fp32 / bf16 MFMAs with random vs zero register operands (power / DVFS probe, round 2):
//   hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe && ./mfma_power_probe
// Every wave keeps 8 independent accumulators busy; 2 waves per SIMD; 256 x 4 workgroups; ~20 ms per measurement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = src[(tid * 8 + k) & 0xffff]; b[k] = src[(tid * 8 + 4 + k) & 0xffff]; }
    float r = 0.f;
    if (MODE == 0) {            // v_mfma_f32_16x16x4_f32
        f32x4 acc[8];
        for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 3], b[(m + 1) & 3], acc[m], 0, 0, 0);
        for (int m = 0; m < 8; ++m) r += acc[m][0] + acc[m][3];
    } else if (MODE == 1) {     // v_mfma_f32_32x32x2_f32
        f32x16 acc[4];
        for (int m = 0; m < 4; ++m) for (int k = 0; k < 16; ++k) acc[m][k] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m & 3], b[(m + 1) & 3], acc[m], 0, 0, 0);
        for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][15];
    } else if (MODE == 2) {     // v_mfma_f32_16x16x32_bf16
        bf16x8 av[2], bv[2];
        for (int k = 0; k < 8; ++k) { av[0][k] = (__bf16)a[k & 3]; av[1][k] = (__bf16)b[k & 3]; bv[0][k] = (__bf16)(a[k & 3] * 0.5f); bv[1][k] = (__bf16)(b[k & 3] * 0.5f); }
        f32x4 acc[8];
        for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m & 1], bv[(m + 1) & 1], acc[m], 0, 0, 0);
        for (int m = 0; m < 8; ++m) r += acc[m][0] + acc[m][3];
    } else {                    // v_mfma_f32_32x32x16_bf16
        bf16x8 av[2], bv[2];
        for (int k = 0; k < 8; ++k) { av[0][k] = (__bf16)a[k & 3]; av[1][k] = (__bf16)b[k & 3]; bv[0][k] = (__bf16)(a[k & 3] * 0.5f); bv[1][k] = (__bf16)(b[k & 3] * 0.5f); }
        f32x16 acc[4];
        for (int m = 0; m < 4; ++m) for (int k = 0; k < 16; ++k) acc[m][k] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m & 1], bv[(m + 1) & 1], acc[m], 0, 0, 0);
        for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][15];
    }
    if (r == 12345.678f) out[tid] = r;
}

template <int MODE>
double run(const float* src, float* out, int iters, double flops_per_mfma, int mfma_per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(256 * 4), dim3(512), 0, 0, src, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256 * 4), dim3(512), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)256 * 4 * 8 * iters * mfma_per_iter * flops_per_mfma;
    return total / (ms * 1e-3) / 1e12;
}

int main() {
    float *src, *out;
    hipMalloc(&src, 65536 * 4); hipMalloc(&out, 256 * 4 * 512 * 4);
    std::vector<float> h(65536);
    const char* names[4] = {"f32 16x16x4 ", "f32 32x32x2 ", "bf16 16x16x32", "bf16 32x32x16"};
    for (int z = 0; z < 2; ++z) {
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = z ? 0.f : ((s >> 8) * (1.f / 8388608.f) - 1.f); }
        hipMemcpy(src, h.data(), 65536 * 4, hipMemcpyHostToDevice);
        double t[4];
        t[0] = run<0>(src, out, 40000, 2.0 * 16 * 16 * 4, 8);
        t[1] = run<1>(src, out, 40000, 2.0 * 32 * 32 * 2, 4);
        t[2] = run<2>(src, out, 40000, 2.0 * 16 * 16 * 32, 8);
        t[3] = run<3>(src, out, 40000, 2.0 * 32 * 32 * 16, 4);
        for (int k = 0; k < 4; ++k) std::printf("%s operands %-6s : %8.1f TF/s\n", names[k], z ? "zero" : "random", t[k]);
    }
    return 0;
}

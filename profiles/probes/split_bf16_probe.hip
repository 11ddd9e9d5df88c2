// fp32 products from split bf16 operands (the open lever for the fp32 headline): accuracy and MFMA-level rate.
//   hipcc --offload-arch=gfx950 -O3 split_bf16_probe.hip -o split_bf16_probe && ./split_bf16_probe
// x = hi + mid + lo exactly (3 x 8 mantissa bits, each part bf16 RNE of the remainder); D = A B from six
// v_mfma_f32_16x16x32_bf16 per operand pair: hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi (the dropped terms are <= 2^-24 relative),
// fp32 accumulation -- against v_mfma_f32_16x16x4_f32 on the same data and a float64 reference on the host.
// Part 1: one wave per 16x16 tile, K = 4000 (the contraction length of a 5^3 conv over 32 channels), 256 tiles, three data
// distributions.  Part 2: register-only streams (random operands, 8 accumulators per wave, 2 waves per SIMD): "fp32" TF/s of both.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;                       // RNE
    const float r1 = x - (float)h;       // exact
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;      // exact
    l = (__bf16)r2;
}

// A [tiles][16][K] row-major, B [tiles][K][16] row-major; D [tiles][16][16]
template <int MODE>      // 0: fp32 MFMA, 1: split bf16 x 6, 2: split bf16 x 9 (all cross terms), 3: plain bf16 (one product)
__global__ void __launch_bounds__(64) gemm_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D, int K) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const float* a = A + (size_t)blockIdx.x * 16 * K;
    const float* b = B + (size_t)blockIdx.x * K * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 4)      // lane (m = i, kk = g) of A, lane (n = i, kk = g) of B
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(size_t)i * K + k + g], b[(size_t)(k + g) * 16 + i], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {   // lane (m = i, k = 8g .. 8g+7)
            bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                split3(a[(size_t)i * K + k + 8 * g + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
                split3(b[(size_t)(k + 8 * g + e) * 16 + i], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
            }
            if (MODE == 3) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0); continue; }
            // small terms first
            if (MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    // lane (n = i, g) holds rows 4g .. 4g+3 of column n
    for (int r = 0; r < 4; ++r) D[(size_t)blockIdx.x * 256 + (4 * g + r) * 16 + i] = acc[r];
}

template <int MODE>      // 0: fp32 16x16x4 stream, 1: six bf16 16x16x32 per k-32 step
__global__ void __launch_bounds__(512) stream(const float* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    float r = 0.f;
    if (MODE == 0) {
        float a[4], b[4];
        for (int k = 0; k < 4; ++k) { a[k] = src[(tid * 8 + k) & 0xffff]; b[k] = src[(tid * 8 + 4 + k) & 0xffff]; }
        f32x4 acc[8];
        for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 3], b[(m + 1) & 3], acc[m], 0, 0, 0);
        for (int m = 0; m < 8; ++m) r += acc[m][0] + acc[m][3];
    } else {
        bf16x8 ap[3], bp[3];
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, l;
            split3(src[(tid * 16 + e) & 0xffff], h, m, l); ap[0][e] = h; ap[1][e] = m; ap[2][e] = l;
            split3(src[(tid * 16 + 8 + e) & 0xffff], h, m, l); bp[0][e] = h; bp[1][e] = m; bp[2][e] = l;
        }
        f32x4 acc[8];
        for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 8; ++m) {       // one accumulator = one output tile: its six products
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[2], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], bp[0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[1], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[0], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[0], acc[m], 0, 0, 0);
            }
        for (int m = 0; m < 8; ++m) r += acc[m][0] + acc[m][3];
    }
    out[tid] = r;
}

int main() {
    const int T = 256, K = 4000;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)T * 16 * K), B((size_t)T * K * 16);
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, (size_t)T * 256 * 4));
    const char* dist_name[3] = {"N(0,1) x N(0,1)", "activations N(1.5,3) x filters 0.05 N(0,1)", "wide dynamic range exp(4 N) x N(0,1)"};
    printf("part 1: D = A B, 256 tiles of 16x16, K = %d; errors against float64, relative to max|D_ref| of the tile (max over tiles) and rel-L2\n", K);
    for (int d = 0; d < 3; ++d) {
        for (auto& v : A) v = d == 0 ? nd(rng) : d == 1 ? 1.5f + 3.f * nd(rng) : std::exp(4.f * nd(rng)) * (nd(rng) > 0 ? 1.f : -1.f);
        for (auto& v : B) v = d == 1 ? 0.05f * nd(rng) : nd(rng);
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> ref((size_t)T * 256);
        for (int t = 0; t < T; ++t)
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double s = 0.0;
                    for (int k = 0; k < K; ++k) s += (double)A[((size_t)t * 16 + m) * K + k] * (double)B[((size_t)t * K + k) * 16 + n];
                    ref[(size_t)t * 256 + m * 16 + n] = s;
                }
        printf("  %s\n", dist_name[d]);
        const char* mode_name[4] = {"v_mfma_f32_16x16x4_f32          ", "split bf16, 6 products          ", "split bf16, 9 products          ", "plain bf16 (1 product)          "};
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(gemm_tile<0>, dim3(T), dim3(64), 0, 0, dA, dB, dD, K);
            else if (mode == 1) hipLaunchKernelGGL(gemm_tile<1>, dim3(T), dim3(64), 0, 0, dA, dB, dD, K);
            else if (mode == 2) hipLaunchKernelGGL(gemm_tile<2>, dim3(T), dim3(64), 0, 0, dA, dB, dD, K);
            else hipLaunchKernelGGL(gemm_tile<3>, dim3(T), dim3(64), 0, 0, dA, dB, dD, K);
            CK(hipDeviceSynchronize());
            std::vector<float> D((size_t)T * 256);
            CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0.0, num = 0.0, den = 0.0;
            for (int t = 0; t < T; ++t) {
                double mx = 0.0, me = 0.0;
                for (int e = 0; e < 256; ++e) {
                    const double r = ref[(size_t)t * 256 + e], x = D[(size_t)t * 256 + e];
                    mx = std::max(mx, std::fabs(r)); me = std::max(me, std::fabs(x - r));
                    num += (x - r) * (x - r); den += r * r;
                }
                worst = std::max(worst, me / mx);
            }
            printf("    %s max err / max|ref| %.3e   rel-L2 %.3e\n", mode_name[mode], worst, std::sqrt(num / den));
        }
    }
    // part 2
    std::vector<float> src(65536);
    for (auto& v : src) v = nd(rng);
    float *dS, *dO;
    CK(hipMalloc(&dS, 65536 * 4)); CK(hipMalloc(&dO, 256 * 4 * 512 * 4));
    CK(hipMemcpy(dS, src.data(), 65536 * 4, hipMemcpyHostToDevice));
    printf("part 2: register-only streams, random operands, 1024 workgroups x 8 waves, 8 accumulators per wave\n");
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = mode == 0 ? 20000 : 4000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(1024), dim3(512), 0, 0, dS, dO, iters);
            else hipLaunchKernelGGL(stream<1>, dim3(1024), dim3(512), 0, 0, dS, dO, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // "fp32" flops: one 16x16 tile x (k = 4 per fp32 MFMA | k = 32 per six bf16 MFMAs)
            const double flops = (double)1024 * 8 * iters * 8 * 2.0 * 16 * 16 * (mode == 0 ? 4 : 32);
            if (rep == 1) printf("    %s %8.3f ms  %.1f TF/s of fp32-equivalent products\n", mode == 0 ? "v_mfma_f32_16x16x4_f32       " : "6 x v_mfma_f32_16x16x32_bf16 ", ms, flops / ms / 1e9);
        }
    }
    return 0;
}

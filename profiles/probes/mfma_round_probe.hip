// How do v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32 round?  (round 6: the f32x3 kernels' error on operands spread over
// 2^+-20 is 1.5x the fp32 MFMA's -- is the bf16 instruction's internal sum truncated / aligned to the largest addend?)
//   hipcc --offload-arch=gfx950 -O2 mfma_round_probe.hip -o mfma_round_probe && ./mfma_round_probe
// One wave, D[0][0] = c + sum_k a_k b_k for hand-picked addends; prints the result against the exact value in units of ulp(result).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// a[32], b[32], c -> out[0] (bf16 MFMA, K = 32), out[1] (fp32 MFMA over k = 0..3 only), out[2]: two chained bf16 MFMAs (k 0..31 then again with a2/b2)
__global__ void __launch_bounds__(64) probe(const float* a, const float* b, float c, float* out) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    bf16x8 A, B;
    for (int e = 0; e < 8; ++e) {
        A[e] = (__bf16)(i == 0 ? a[8 * g + e] : 0.f);
        B[e] = (__bf16)(i == 0 ? b[8 * g + e] : 0.f);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (lane == 0) acc[0] = c;
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc, 0, 0, 0);
    if (lane == 0) out[0] = d[0];
    const float af = i == 0 ? a[g] : 0.f, bf = i == 0 ? b[g] : 0.f;
    f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc, 0, 0, 0);
    if (lane == 0) out[1] = d2[0];
}

static float run(const float* a, const float* b, float c, float* d_a, float* d_b, float* d_o, float* fp32_out) {
    CK(hipMemcpy(d_a, a, 32 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b, 32 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_b, c, d_o);
    float o[2];
    CK(hipMemcpy(o, d_o, 8, hipMemcpyDeviceToHost));
    *fp32_out = o[1];
    return o[0];
}

int main() {
    float *d_a, *d_b, *d_o;
    CK(hipMalloc(&d_a, 128)); CK(hipMalloc(&d_b, 128)); CK(hipMalloc(&d_o, 16));
    struct Case { const char* name; float c; float a[32]; float b[32]; };
    auto report = [&](const char* name, float c, const float* a, const float* b) {
        long double exact = c, exact4 = c;
        for (int k = 0; k < 32; ++k) { exact += (long double)a[k] * b[k]; if (k < 4) exact4 += (long double)a[k] * b[k]; }
        float f32;
        const float got = run(a, b, c, d_a, d_b, d_o, &f32);
        const float r = (float)exact, r4 = (float)exact4;
        const double u = std::ldexp(1.0, std::ilogb((double)(r != 0 ? r : 1)) - 23), u4 = std::ldexp(1.0, std::ilogb((double)(r4 != 0 ? r4 : 1)) - 23);
        printf("%-58s bf16x32: got %.9g exact %.12Lg  err %+.3f ulp | f32x4 (k<4): got %.9g exact %.12Lg err %+.3f ulp\n", name, got, exact,
               (double)((long double)got - exact) / u, f32, exact4, (double)((long double)f32 - exact4) / u4);
    };
    float a[32], b[32];
    auto clear = [&]() { for (int k = 0; k < 32; ++k) { a[k] = 0; b[k] = 0; } };
    const float U = std::ldexp(1.f, -23);      // ulp(1.0)
    clear(); a[0] = 1.5f; b[0] = 0.5f * U;                    report("c=1 + 0.75 ulp (RNE: +1, RTZ: 0)", 1.f, a, b);
    clear(); a[0] = 1.0f; b[0] = 0.5f * U;                    report("c=1 + 0.5 ulp tie (RNE-even: 0)", 1.f, a, b);
    clear(); a[0] = 1.5f; b[0] = U;                           report("c=1 + 1.5 ulp tie (RNE-even: +2)", 1.f, a, b);
    clear(); a[0] = -1.0f; b[0] = 0.125f * U;                 report("c=1 - 0.125 ulp(1) = -0.25 ulp below (RNE: 0, RTZ: -1)", 1.f, a, b);
    clear(); for (int k = 0; k < 32; ++k) { a[k] = 1.f; b[k] = 0.25f * U; } report("c=1 + 32 x 0.25 ulp (exact: +8 ulp)", 1.f, a, b);
    clear(); for (int k = 0; k < 4; ++k) { a[k] = 1.f; b[k] = 0.25f * U; } report("c=1 + 4 x 0.25 ulp (exact: +1 ulp)", 1.f, a, b);
    clear(); a[0] = 1.f; b[0] = 1.f; for (int k = 1; k < 32; ++k) { a[k] = 1.5f; b[k] = 0.25f * U; } report("c=0, p0=1 + 31 x 0.375 ulp (exact 11.625 ulp)", 0.f, a, b);
    clear(); a[0] = 1.f; b[0] = 1.f; for (int k = 1; k < 4; ++k) { a[k] = 1.5f; b[k] = 0.25f * U; } report("c=0, p0=1 + 3 x 0.375 ulp (exact 1.125 ulp)", 0.f, a, b);
    // floor or truncation toward zero?  small NEGATIVE products next to a large positive one, and the mirror image
    clear(); a[0] = 1.f; b[0] = 1.f; for (int k = 1; k < 4; ++k) { a[k] = -1.5f; b[k] = 0.25f * U; } report("c=0, p0=+1, 3 x -0.375 ulp (RTZ: +1.125 err, floor: -0.375)", 0.f, a, b);
    clear(); a[0] = -1.f; b[0] = 1.f; for (int k = 1; k < 4; ++k) { a[k] = 1.5f; b[k] = 0.25f * U; } report("c=0, p0=-1, 3 x +0.375 ulp (RTZ: -1.125 err, floor: -1.125)", 0.f, a, b);
    clear(); a[0] = -1.f; b[0] = 1.f; for (int k = 1; k < 4; ++k) { a[k] = -1.5f; b[k] = 0.25f * U; } report("c=0, p0=-1, 3 x -0.375 ulp (RTZ: +1.125 err, floor: -0.375)", 0.f, a, b);
    clear(); a[0] = 1.f; b[0] = 1.f; for (int k = 1; k < 8; ++k) { a[k] = -1.25f; b[k] = 0.125f * U; } report("c=0, p0=+1, 7 x -0.156 ulp (floor: each -> -0.5 ulp)", 0.f, a, b);
    for (int sh = 20; sh <= 48; sh += 4) {
        char nm[96]; snprintf(nm, sizeof nm, "c=0: 2^%d - 2^%d + 30 x 1.0 (internal width: exact 30)", sh, sh);
        clear(); a[0] = std::ldexp(1.f, sh / 2); b[0] = std::ldexp(1.f, sh - sh / 2); a[1] = -a[0]; b[1] = b[0];
        for (int k = 2; k < 32; ++k) { a[k] = 1.f; b[k] = 1.f; }
        report(nm, 0.f, a, b);
    }
    for (int sh = 20; sh <= 48; sh += 4) {
        char nm[96]; snprintf(nm, sizeof nm, "c=2^%d, p0 = -2^%d, + 2 x 1.0 (exact 2; k<4 form)", sh, sh);
        clear(); a[0] = -std::ldexp(1.f, sh / 2); b[0] = std::ldexp(1.f, sh - sh / 2);
        a[1] = a[2] = 1.f; b[1] = b[2] = 1.f;
        report(nm, std::ldexp(1.f, sh), a, b);
    }
    // one big product and small ones just below the cut: how many bits below the largest addend survive?
    for (int sh = 22; sh <= 34; sh += 2) {
        char nm[96]; snprintf(nm, sizeof nm, "c=0: p0 = 1, p1..3 = 2^-%d each (k<4 form)", sh);
        clear(); a[0] = 1.f; b[0] = 1.f; for (int k = 1; k < 4; ++k) { a[k] = 1.f; b[k] = std::ldexp(1.f, -sh); }
        report(nm, 0.f, a, b);
    }
    return 0;
}

// conv2_b16.hip -- the 2x2x2 stride-2 pair of the V-Net (reference layers2.py:78-94: down_convolution / up_convolution and, each being
// the other's backward-data, model.py:660) on bf16 tensors WITHOUT an LDS tile: these launches are HBM-bound (AI 12.8 flop/B at
// level 1), every input element is used by exactly one output voxel's GEMV, so there is nothing to stage -- a lane's MFMA operand
// is ONE 16-byte global load of the NDHWC tensor, a lane's result ONE 16-byte store.
//
// Both kernels see the filter as W[tap = (a,b,c)][Cf fine channels][Cc coarse channels] fp32 -- the memory layout of BOTH TF filters
// (down: [2,2,2,Cin=Cf,Cout=Cc], transposed: [2,2,2,Cout=Cf,Cin=Cc]) -- round it to bf16 (RNE) and keep it in LDS in fragment order:
//   DOWN  coarse[v][cc]        = sum_tap sum_cf fine[2v + tap][cf] W[tap][cf][cc]     (down conv forward, transposed conv backward-data)
//   UP    fine[2v + tap][cf]  (+)= sum_cc coarse[v][cc] W[tap][cf][cc]                 (transposed conv forward, down conv backward-data)
// v_mfma_f32_16x16x32_bf16, D[16 output channels][16 voxels along x]; the output-channel order inside the MFMA blocks is permuted
// (free: it is only the order of the filter rows in LDS) so that a lane ends up with 8 CONSECUTIVE channels of one voxel.
// Widths: Cf a multiple of 16, Cc a multiple of 32, 8 * Cf * Cc * 2 bytes <= 64 KB (V-Net levels 1 and 2); other shapes take the
// generic kernels (vnet_conv2_fwd_b16).  Same arithmetic as those: exact bf16 x bf16 products, fp32 accumulation, one rounding.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) u32x4* gvec16_t;
__device__ __attribute__((aligned(64))) const unsigned int conv2_zero_line[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
__device__ __forceinline__ u32x4 ld16z(const unsigned short* p, bool ok) {      // address select, not a data select (conv_kernels.h)
    gvec16_t src = ok ? (gvec16_t)(p) : (gvec16_t)(conv2_zero_line);
    return *src;
}
__device__ __forceinline__ float blo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

struct C2Args {
    const unsigned short* in; unsigned short* out; const float* w; const float* bias; float* stats;
    int Cf, Cc, B, Df, Hf, Wf, Dc, Hc, Wc;      // fine / coarse spatial dims
    int nseg, segx;                              // segments of 16 coarse voxels along x: nseg = B * Dc * Hc * segx
    int accum;
};

constexpr int C2_THREADS = 256;

// ---- DOWN: CF fine channels -> CC coarse channels --------------------------------------------------------------------------
template <int CF, int CC, bool STATS>
__global__ void __launch_bounds__(C2_THREADS) conv2_down_b16_kernel(C2Args a) {
    constexpr int S = 2 * CF / 32, NB = CC / 16, NPAIR = NB / 2;
    static_assert(CF % 16 == 0 && CC % 32 == 0, "widths");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(smem);                       // [ab 4][s][n][lane 64] fragments of 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    // filter -> LDS: fragment (ab, s, n), lane (r = row, gg): 8 consecutive k = 32 s + 8 gg .. of row r
    for (int u = tid; u < 4 * S * NB * 64; u += C2_THREADS) {
        const int l = u & 63, n = (u >> 6) % NB, s = ((u >> 6) / NB) % S, ab = (u >> 6) / (NB * S);
        const int r = l & 15, gg = l >> 4;
        const int cc = 32 * (n >> 1) + 8 * (r >> 2) + 4 * (n & 1) + (r & 3);
        const int kk = 32 * s + 8 * gg, c = kk / CF, cf0 = kk % CF;
        const float* src = a.w + ((size_t)((ab * 2 + c) * CF + cf0)) * CC + cc;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = src[(size_t)e * CC];
        const u32x4 pk = {pk_bf16(f[0], f[1]), pk_bf16(f[2], f[3]), pk_bf16(f[4], f[5]), pk_bf16(f[6], f[7])};
        wl[u] = pk;
    }
    __syncthreads();
    float bia[NPAIR][8];
#pragma unroll
    for (int m = 0; m < NPAIR; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) bia[m][e] = a.bias ? a.bias[32 * m + 8 * g + e] : 0.f;
    float s1[STATS ? NPAIR : 1][8], s2[STATS ? NPAIR : 1][8];
#pragma unroll
    for (int m = 0; m < (STATS ? NPAIR : 1); ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[m][e] = s2[m][e] = 0.f;

    const int nwaves = gridDim.x * (C2_THREADS / 64), w0 = blockIdx.x * (C2_THREADS / 64) + wave;
    u32x4 xb[4][S];
    auto issue = [&](int seg) {
        const int sx = seg % a.segx; int t = seg / a.segx;
        const int yc = t % a.Hc; t /= a.Hc;
        const int zc = t % a.Dc, b = t / a.Dc;
        const int xc = sx * 16 + i;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const int zf = 2 * zc + (ab >> 1), yf = 2 * yc + (ab & 1);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int kk = 32 * s + 8 * g, c = kk / CF, cf0 = kk % CF;
                const int xf = 2 * xc + c;
                const bool ok = xc < a.Wc && zf < a.Df && yf < a.Hf && xf < a.Wf;
                const size_t fv = ok ? ((size_t)(b * a.Df + zf) * a.Hf + yf) * a.Wf + xf : 0;
                xb[ab][s] = ld16z(a.in + fv * CF + cf0, ok);
            }
        }
    };
    if (w0 < a.nseg) issue(w0);
    for (int seg = w0; seg < a.nseg; seg += nwaves) {
        bf16x8 xf[4][S];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int s = 0; s < S; ++s) xf[ab][s] = __builtin_bit_cast(bf16x8, xb[ab][s]);
        if (seg + nwaves < a.nseg) issue(seg + nwaves);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, wl[((ab * S + s) * NB + n) * 64 + lane]);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[ab][s], acc[n], 0, 0, 0);
                }
        const int sx = seg % a.segx; int t = seg / a.segx;
        const int yc = t % a.Hc; t /= a.Hc;
        const int zc = t % a.Dc, b = t / a.Dc;
        const int xc = sx * 16 + i;
        if (xc < a.Wc) {
            const size_t cv = ((size_t)(b * a.Dc + zc) * a.Hc + yc) * a.Wc + xc;
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                float e[8] = {acc[2 * m][0], acc[2 * m][1], acc[2 * m][2], acc[2 * m][3], acc[2 * m + 1][0], acc[2 * m + 1][1], acc[2 * m + 1][2], acc[2 * m + 1][3]};
                u32x4* dst = reinterpret_cast<u32x4*>(a.out + cv * CC + 32 * m + 8 * g);
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] += bia[m][k];
                if (a.accum) {
                    const u32x4 o = *dst;
                    e[0] += blo(o[0]); e[1] += bhi(o[0]); e[2] += blo(o[1]); e[3] += bhi(o[1]);
                    e[4] += blo(o[2]); e[5] += bhi(o[2]); e[6] += blo(o[3]); e[7] += bhi(o[3]);
                }
                const u32x4 pk = {pk_bf16(e[0], e[1]), pk_bf16(e[2], e[3]), pk_bf16(e[4], e[5]), pk_bf16(e[6], e[7])};
                if constexpr (STATS) {              // statistics of the ROUNDED values (what the batch-norm behind normalises)
                    const float v[8] = {blo(pk[0]), bhi(pk[0]), blo(pk[1]), bhi(pk[1]), blo(pk[2]), bhi(pk[2]), blo(pk[3]), bhi(pk[3])};
#pragma unroll
                    for (int k = 0; k < 8; ++k) { s1[m][k] += v[k]; s2[m][k] += v[k] * v[k]; }
                }
                *dst = pk;
            }
        }
    }
    if constexpr (STATS) {
        // lanes of one g hold the same channels for 16 voxels: butterfly over i, then across the four waves through LDS
        float* red = reinterpret_cast<float*>(smem) + (4 * S * NB * 64 * 16) / 4;       // behind the filter image
#pragma unroll
        for (int m = 0; m < NPAIR; ++m)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { s1[m][k] += __shfl_xor(s1[m][k], off, 64); s2[m][k] += __shfl_xor(s2[m][k], off, 64); }
                if (i == 0) { red[wave * 2 * CC + 32 * m + 8 * g + k] = s1[m][k]; red[wave * 2 * CC + CC + 32 * m + 8 * g + k] = s2[m][k]; }
            }
        __syncthreads();
        for (int q = tid; q < 2 * CC; q += C2_THREADS)
            a.stats[(size_t)blockIdx.x * 2 * CC + q] = red[q] + red[2 * CC + q] + red[4 * CC + q] + red[6 * CC + q];
    }
}

// ---- UP: CC coarse channels -> CF fine channels at the 8 fine positions of every coarse voxel -------------------------------
template <int CF, int CC>
__global__ void __launch_bounds__(C2_THREADS) conv2_up_b16_kernel(C2Args a) {
    constexpr int S = CC / 32, NB = 2 * CF / 16, NPAIR = NB / 2;         // per (a,b): rows = (c, cf)
    static_assert(CF % 16 == 0 && CC % 32 == 0, "widths");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(smem);                           // [ab 4][s][n][lane 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    for (int u = tid; u < 4 * S * NB * 64; u += C2_THREADS) {
        const int l = u & 63, n = (u >> 6) % NB, s = ((u >> 6) / NB) % S, ab = (u >> 6) / (NB * S);
        const int r = l & 15, gg = l >> 4;
        // row r of block n of pair m = n / 2  ->  (c, cf): the lane that ends up with rows 4 g' .. 4 g' + 3 (g' = r / 4) holds
        // c = g' / 2 and the channel octet 2 m + (g' & 1)
        const int gp = r >> 2, c = gp >> 1, cf = 8 * (2 * (n >> 1) + (gp & 1)) + 4 * (n & 1) + (r & 3);
        const float* src = a.w + ((size_t)((ab * 2 + c) * CF + cf)) * CC + 32 * s + 8 * gg;
        const u32x4 pk = {pk_bf16(src[0], src[1]), pk_bf16(src[2], src[3]), pk_bf16(src[4], src[5]), pk_bf16(src[6], src[7])};
        wl[u] = pk;
    }
    __syncthreads();
    float bia[NPAIR][8];
#pragma unroll
    for (int m = 0; m < NPAIR; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) bia[m][e] = a.bias ? a.bias[8 * (2 * m + (g & 1)) + e] : 0.f;

    const int nwaves = gridDim.x * (C2_THREADS / 64), w0 = blockIdx.x * (C2_THREADS / 64) + wave;
    u32x4 xb[S];
    auto issue = [&](int seg) {
        const int sx = seg % a.segx; const int row = seg / a.segx;                 // row = (b, zc, yc) flattened
        const int xc = sx * 16 + i;
        const bool ok = xc < a.Wc;
#pragma unroll
        for (int s = 0; s < S; ++s) xb[s] = ld16z(a.in + ((size_t)row * a.Wc + (ok ? xc : 0)) * CC + 32 * s + 8 * g, ok);
    };
    if (w0 < a.nseg) issue(w0);
    for (int seg = w0; seg < a.nseg; seg += nwaves) {
        bf16x8 xf[S];
#pragma unroll
        for (int s = 0; s < S; ++s) xf[s] = __builtin_bit_cast(bf16x8, xb[s]);
        const int sx = seg % a.segx; int t = seg / a.segx;
        const int yc = t % a.Hc; t /= a.Hc;
        const int zc = t % a.Dc, b = t / a.Dc;
        const int xc = sx * 16 + i;
        if (seg + nwaves < a.nseg) issue(seg + nwaves);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            f32x4 acc[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, wl[((ab * S + s) * NB + n) * 64 + lane]);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[s], acc[n], 0, 0, 0);
                }
            const int zf = 2 * zc + (ab >> 1), yf = 2 * yc + (ab & 1), xf_ = 2 * xc + (g >> 1);
            if (xc < a.Wc && zf < a.Df && yf < a.Hf && xf_ < a.Wf) {
                const size_t fv = ((size_t)(b * a.Df + zf) * a.Hf + yf) * a.Wf + xf_;
#pragma unroll
                for (int m = 0; m < NPAIR; ++m) {
                    float e[8] = {acc[2 * m][0], acc[2 * m][1], acc[2 * m][2], acc[2 * m][3], acc[2 * m + 1][0], acc[2 * m + 1][1], acc[2 * m + 1][2], acc[2 * m + 1][3]};
                    u32x4* dst = reinterpret_cast<u32x4*>(a.out + fv * CF + 8 * (2 * m + (g & 1)));
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] += bia[m][k];
                    if (a.accum) {
                        const u32x4 o = *dst;
                        e[0] += blo(o[0]); e[1] += bhi(o[0]); e[2] += blo(o[1]); e[3] += bhi(o[1]);
                        e[4] += blo(o[2]); e[5] += bhi(o[2]); e[6] += blo(o[3]); e[7] += bhi(o[3]);
                    }
                    const u32x4 pk = {pk_bf16(e[0], e[1]), pk_bf16(e[2], e[3]), pk_bf16(e[4], e[5]), pk_bf16(e[6], e[7])};
                    *dst = pk;
                }
            }
        }
    }
}

// =====================================================================================================================
// fp32 twins (the reference's arithmetic): v_mfma_f32_16x16x4_f32, exact fp32.  A lane's operand is one 16-byte load = 4 consecutive
// channels; the four MFMA steps of a "k quad" take elements x, y, z, w of both fragments (hardware k index g <-> channel 4 g + j: a
// K permutation shared by both operands, as in conv_kernels.h).  Same decomposition, same permuted output-channel order (a lane
// ends up with 8 consecutive channels = two adjacent 16-byte stores).  The fp32 filter image needs 16 KB (level 1) / 64 KB (level 2).
// =====================================================================================================================
struct C2ArgsF {
    const float* in; float* out; const float* w; const float* bias; float* stats;
    int Cf, Cc, B, Df, Hf, Wf, Dc, Hc, Wc;
    int nseg, segx;
    int accum;
};

__device__ __forceinline__ float4 ld16zf(const float* p, bool ok) {
    typedef const __attribute__((address_space(1))) f32x4* gf4_t;
    gf4_t src = ok ? (gf4_t)(p) : (gf4_t)(conv2_zero_line);
    const f32x4 v = *src;
    return make_float4(v[0], v[1], v[2], v[3]);
}

template <int CF, int CC, bool STATS>
__global__ void __launch_bounds__(C2_THREADS) conv2_down_f32_kernel(C2ArgsF a) {
    constexpr int NQ = 2 * CF / 16, NB = CC / 16, NPAIR = NB / 2;        // k quads per (a,b): 16 channels each
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* wl = reinterpret_cast<float4*>(smem);                       // [ab 4][q][n][lane 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    for (int u = tid; u < 4 * NQ * NB * 64; u += C2_THREADS) {
        const int l = u & 63, n = (u >> 6) % NB, q = ((u >> 6) / NB) % NQ, ab = (u >> 6) / (NB * NQ);
        const int r = l & 15, gg = l >> 4;
        const int cc = 32 * (n >> 1) + 8 * (r >> 2) + 4 * (n & 1) + (r & 3);
        const int kk = 16 * q + 4 * gg, c = kk / CF, cf0 = kk % CF;
        const float* src = a.w + ((size_t)((ab * 2 + c) * CF + cf0)) * CC + cc;
        wl[u] = make_float4(src[0], src[(size_t)CC], src[(size_t)2 * CC], src[(size_t)3 * CC]);
    }
    __syncthreads();
    float bia[NPAIR][8];
#pragma unroll
    for (int m = 0; m < NPAIR; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) bia[m][e] = a.bias ? a.bias[32 * m + 8 * g + e] : 0.f;
    float s1[STATS ? NPAIR : 1][8], s2[STATS ? NPAIR : 1][8];
#pragma unroll
    for (int m = 0; m < (STATS ? NPAIR : 1); ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[m][e] = s2[m][e] = 0.f;

    const int nwaves = gridDim.x * (C2_THREADS / 64), w0 = blockIdx.x * (C2_THREADS / 64) + wave;
    float4 xb[4][NQ];
    auto issue = [&](int seg) {
        const int sx = seg % a.segx; int t = seg / a.segx;
        const int yc = t % a.Hc; t /= a.Hc;
        const int zc = t % a.Dc, b = t / a.Dc;
        const int xc = sx * 16 + i;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const int zf = 2 * zc + (ab >> 1), yf = 2 * yc + (ab & 1);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int kk = 16 * q + 4 * g, c = kk / CF, cf0 = kk % CF;
                const int xf = 2 * xc + c;
                const bool ok = xc < a.Wc && zf < a.Df && yf < a.Hf && xf < a.Wf;
                const size_t fv = ok ? ((size_t)(b * a.Df + zf) * a.Hf + yf) * a.Wf + xf : 0;
                xb[ab][q] = ld16zf(a.in + fv * CF + cf0, ok);
            }
        }
    };
    if (w0 < a.nseg) issue(w0);
    for (int seg = w0; seg < a.nseg; seg += nwaves) {
        float4 xf[4][NQ];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int q = 0; q < NQ; ++q) xf[ab][q] = xb[ab][q];
        if (seg + nwaves < a.nseg) issue(seg + nwaves);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const float4 af = wl[((ab * NQ + q) * NB + n) * 64 + lane];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, xf[ab][q].x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, xf[ab][q].y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, xf[ab][q].z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, xf[ab][q].w, acc[n], 0, 0, 0);
                }
        const int sx = seg % a.segx; int t = seg / a.segx;
        const int yc = t % a.Hc; t /= a.Hc;
        const int zc = t % a.Dc, b = t / a.Dc;
        const int xc = sx * 16 + i;
        if (xc < a.Wc) {
            const size_t cv = ((size_t)(b * a.Dc + zc) * a.Hc + yc) * a.Wc + xc;
#pragma unroll
            for (int m = 0; m < NPAIR; ++m) {
                float e[8] = {acc[2 * m][0], acc[2 * m][1], acc[2 * m][2], acc[2 * m][3], acc[2 * m + 1][0], acc[2 * m + 1][1], acc[2 * m + 1][2], acc[2 * m + 1][3]};
                float4* dst = reinterpret_cast<float4*>(a.out + cv * CC + 32 * m + 8 * g);
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] += bia[m][k];
                if (a.accum) {
                    const float4 o0 = dst[0], o1 = dst[1];
                    e[0] += o0.x; e[1] += o0.y; e[2] += o0.z; e[3] += o0.w; e[4] += o1.x; e[5] += o1.y; e[6] += o1.z; e[7] += o1.w;
                }
                if constexpr (STATS) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { s1[m][k] += e[k]; s2[m][k] += e[k] * e[k]; }
                }
                dst[0] = make_float4(e[0], e[1], e[2], e[3]);
                dst[1] = make_float4(e[4], e[5], e[6], e[7]);
            }
        }
    }
    if constexpr (STATS) {
        float* red = reinterpret_cast<float*>(smem) + 4 * NQ * NB * 64 * 4;       // behind the filter image
#pragma unroll
        for (int m = 0; m < NPAIR; ++m)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { s1[m][k] += __shfl_xor(s1[m][k], off, 64); s2[m][k] += __shfl_xor(s2[m][k], off, 64); }
                if (i == 0) { red[wave * 2 * CC + 32 * m + 8 * g + k] = s1[m][k]; red[wave * 2 * CC + CC + 32 * m + 8 * g + k] = s2[m][k]; }
            }
        __syncthreads();
        for (int q = tid; q < 2 * CC; q += C2_THREADS)
            a.stats[(size_t)blockIdx.x * 2 * CC + q] = red[q] + red[2 * CC + q] + red[4 * CC + q] + red[6 * CC + q];
    }
}

template <int CF, int CC>
__global__ void __launch_bounds__(C2_THREADS) conv2_up_f32_kernel(C2ArgsF a) {
    constexpr int NQ = CC / 16, NB = 2 * CF / 16, NPAIR = NB / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* wl = reinterpret_cast<float4*>(smem);                           // [ab 4][q][n][lane 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    for (int u = tid; u < 4 * NQ * NB * 64; u += C2_THREADS) {
        const int l = u & 63, n = (u >> 6) % NB, q = ((u >> 6) / NB) % NQ, ab = (u >> 6) / (NB * NQ);
        const int r = l & 15, gg = l >> 4;
        const int gp = r >> 2, c = gp >> 1, cf = 8 * (2 * (n >> 1) + (gp & 1)) + 4 * (n & 1) + (r & 3);
        wl[u] = *reinterpret_cast<const float4*>(a.w + ((size_t)((ab * 2 + c) * CF + cf)) * CC + 16 * q + 4 * gg);
    }
    __syncthreads();
    float bia[NPAIR][8];
#pragma unroll
    for (int m = 0; m < NPAIR; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) bia[m][e] = a.bias ? a.bias[8 * (2 * m + (g & 1)) + e] : 0.f;

    const int nwaves = gridDim.x * (C2_THREADS / 64), w0 = blockIdx.x * (C2_THREADS / 64) + wave;
    float4 xb[NQ];
    auto issue = [&](int seg) {
        const int sx = seg % a.segx; const int row = seg / a.segx;
        const int xc = sx * 16 + i;
        const bool ok = xc < a.Wc;
#pragma unroll
        for (int q = 0; q < NQ; ++q) xb[q] = ld16zf(a.in + ((size_t)row * a.Wc + (ok ? xc : 0)) * CC + 16 * q + 4 * g, ok);
    };
    if (w0 < a.nseg) issue(w0);
    for (int seg = w0; seg < a.nseg; seg += nwaves) {
        float4 xf[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) xf[q] = xb[q];
        const int sx = seg % a.segx; int t = seg / a.segx;
        const int yc = t % a.Hc; t /= a.Hc;
        const int zc = t % a.Dc, b = t / a.Dc;
        const int xc = sx * 16 + i;
        if (seg + nwaves < a.nseg) issue(seg + nwaves);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            f32x4 acc[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const float4 af = wl[((ab * NQ + q) * NB + n) * 64 + lane];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, xf[q].x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, xf[q].y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, xf[q].z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, xf[q].w, acc[n], 0, 0, 0);
                }
            const int zf = 2 * zc + (ab >> 1), yf = 2 * yc + (ab & 1), xf_ = 2 * xc + (g >> 1);
            if (xc < a.Wc && zf < a.Df && yf < a.Hf && xf_ < a.Wf) {
                const size_t fv = ((size_t)(b * a.Df + zf) * a.Hf + yf) * a.Wf + xf_;
#pragma unroll
                for (int m = 0; m < NPAIR; ++m) {
                    float e[8] = {acc[2 * m][0], acc[2 * m][1], acc[2 * m][2], acc[2 * m][3], acc[2 * m + 1][0], acc[2 * m + 1][1], acc[2 * m + 1][2], acc[2 * m + 1][3]};
                    float4* dst = reinterpret_cast<float4*>(a.out + fv * CF + 8 * (2 * m + (g & 1)));
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] += bia[m][k];
                    if (a.accum) {
                        const float4 o0 = dst[0], o1 = dst[1];
                        e[0] += o0.x; e[1] += o0.y; e[2] += o0.z; e[3] += o0.w; e[4] += o1.x; e[5] += o1.y; e[6] += o1.z; e[7] += o1.w;
                    }
                    dst[0] = make_float4(e[0], e[1], e[2], e[3]);
                    dst[1] = make_float4(e[4], e[5], e[6], e[7]);
                }
            }
        }
    }
}

template <typename K>
int c2_launch_f(K kernel, const C2ArgsF& a, int grid, size_t lds, hipStream_t st, unsigned long long& done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit)) {        // the level-2 filter image (64 KB) + statistics scratch exceeds the default dynamic-LDS limit
        const int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e) return e;
        __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(C2_THREADS), lds, st, a);
    return (int)hipGetLastError();
}

// exactly the instantiated pairs (levels 1 and 2 of the V-Net: Cc = 2 Cf); other widths take the generic kernels
inline bool c2_widths_ok(int Cf, int Cc) { return (Cf == 16 && Cc == 32) || (Cf == 32 && Cc == 64); }

inline int c2_grid(int nseg) {
    // >= 2 segments per wave where the problem allows, at most 2048 workgroups (8 per CU, 32 waves: full occupancy)
    int wg = (nseg + 7) / 8;
    if (wg > 2048) wg = 2048;
    if (wg < 1) wg = 1;
    return wg;
}

template <typename K>
int c2_launch(K kernel, const C2Args& a, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(C2_THREADS), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int vnet_conv2_direct_ok(int Cf, int Cc) { return c2_widths_ok(Cf, Cc) ? 1 : 0; }

// partial-statistics rows of the DOWN kernel (one per workgroup), 0 if these widths take the generic kernel
int vnet_conv2_direct_stats_rows(int Cf, int Cc, int B, int Dc, int Hc, int Wc) {
    if (!c2_widths_ok(Cf, Cc) || B <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0) return 0;
    return c2_grid(B * Dc * Hc * ((Wc + 15) / 16));
}

// down = 1: coarse[B,Dc,Hc,Wc,Cc] = conv2x2x2_stride2(fine[B,Df,Hf,Wf,Cf], w) (+ bias[Cc]);  statistics rows: above
// down = 0: fine[B,Df,Hf,Wf,Cf] (+)= conv2x2x2_transposed(coarse[B,Dc,Hc,Wc,Cc], w) (+ bias[Cf])
// w: fp32 [8][Cf][Cc] (the TF layout of either filter).  accum: add onto the stored output (one rounding of the sum).
int vnet_conv2_direct_b16(int down, const void* in, void* out, const float* w, const float* bias, int Cf, int Cc,
                          int B, int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int accum, float* stats, void* stream) {
    if (!in || !out || !w || B <= 0 || Df <= 0 || Hf <= 0 || Wf <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0) return VNET_E_BADARG;
    if (!c2_widths_ok(Cf, Cc) || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15)) return VNET_E_UNSUPPORTED;
    if (Dc != (Df + 1) / 2 || Hc != (Hf + 1) / 2 || Wc != (Wf + 1) / 2) return VNET_E_BADARG;
    if (stats && (!down || accum)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    C2Args a{};
    a.in = reinterpret_cast<const unsigned short*>(in); a.out = reinterpret_cast<unsigned short*>(out); a.w = w; a.bias = bias; a.stats = stats;
    a.Cf = Cf; a.Cc = Cc; a.B = B; a.Df = Df; a.Hf = Hf; a.Wf = Wf; a.Dc = Dc; a.Hc = Hc; a.Wc = Wc;
    a.segx = (Wc + 15) / 16; a.nseg = B * Dc * Hc * a.segx; a.accum = accum ? 1 : 0;
    const int grid = c2_grid(a.nseg);
    const size_t wbytes = (size_t)8 * Cf * Cc * 2;
    int e;
    if (down) {
        const size_t lds = wbytes + (stats ? (size_t)4 * 2 * Cc * sizeof(float) : 0);
        if (Cf == 16 && Cc == 32) e = stats ? c2_launch(conv2_down_b16_kernel<16, 32, true>, a, grid, lds, st) : c2_launch(conv2_down_b16_kernel<16, 32, false>, a, grid, lds, st);
        else if (Cf == 32 && Cc == 64) e = stats ? c2_launch(conv2_down_b16_kernel<32, 64, true>, a, grid, lds, st) : c2_launch(conv2_down_b16_kernel<32, 64, false>, a, grid, lds, st);
        else return VNET_E_UNSUPPORTED;
    } else {
        if (Cf == 16 && Cc == 32) e = c2_launch(conv2_up_b16_kernel<16, 32>, a, grid, wbytes, st);
        else if (Cf == 32 && Cc == 64) e = c2_launch(conv2_up_b16_kernel<32, 64>, a, grid, wbytes, st);
        else return VNET_E_UNSUPPORTED;
    }
    return e;
}

// fp32 tensors (the reference's arithmetic); arguments as vnet_conv2_direct_b16, float pointers
int vnet_conv2_direct_f32(int down, const float* in, float* out, const float* w, const float* bias, int Cf, int Cc,
                          int B, int Df, int Hf, int Wf, int Dc, int Hc, int Wc, int accum, float* stats, void* stream) {
    if (!in || !out || !w || B <= 0 || Df <= 0 || Hf <= 0 || Wf <= 0 || Dc <= 0 || Hc <= 0 || Wc <= 0) return VNET_E_BADARG;
    if (!c2_widths_ok(Cf, Cc) || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w)) & 15)) return VNET_E_UNSUPPORTED;
    if (Dc != (Df + 1) / 2 || Hc != (Hf + 1) / 2 || Wc != (Wf + 1) / 2) return VNET_E_BADARG;
    if (stats && (!down || accum)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    C2ArgsF a{};
    a.in = in; a.out = out; a.w = w; a.bias = bias; a.stats = stats;
    a.Cf = Cf; a.Cc = Cc; a.B = B; a.Df = Df; a.Hf = Hf; a.Wf = Wf; a.Dc = Dc; a.Hc = Hc; a.Wc = Wc;
    a.segx = (Wc + 15) / 16; a.nseg = B * Dc * Hc * a.segx; a.accum = accum ? 1 : 0;
    const int grid = c2_grid(a.nseg);
    const size_t wbytes = (size_t)8 * Cf * Cc * 4;
    const size_t lds = wbytes + (size_t)4 * 2 * Cc * sizeof(float);
    static unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
    if (down) {
        if (Cf == 16 && Cc == 32) return stats ? c2_launch_f(conv2_down_f32_kernel<16, 32, true>, a, grid, lds, st, m0) : c2_launch_f(conv2_down_f32_kernel<16, 32, false>, a, grid, lds, st, m1);
        if (Cf == 32 && Cc == 64) return stats ? c2_launch_f(conv2_down_f32_kernel<32, 64, true>, a, grid, lds, st, m2) : c2_launch_f(conv2_down_f32_kernel<32, 64, false>, a, grid, lds, st, m3);
        return VNET_E_UNSUPPORTED;
    }
    if (Cf == 16 && Cc == 32) return c2_launch_f(conv2_up_f32_kernel<16, 32>, a, grid, lds, st, m4);
    if (Cf == 32 && Cc == 64) return c2_launch_f(conv2_up_f32_kernel<32, 64>, a, grid, lds, st, m5);
    return VNET_E_UNSUPPORTED;
}

}  // extern "C"

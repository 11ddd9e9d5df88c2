// x3_pack.h -- filter images of the f32x3 convolution kernels (conv_x3.h): included by conv_kernels.h inside its anonymous
// namespace (after bf_lo / bf_hi), so that the batched repack (pack_batched_kernel) can emit them too.
#pragma once
constexpr int X3_NPAIR = 63;             // tap pairs per 16-channel chunk (125 taps: 62 pairs + one single)

// exact three-way split of four fp32 values into packed bf16 pairs
__device__ __forceinline__ void x3_split4(const float4 v, u32x2& h, u32x2& m, u32x2& l) {
    h = u32x2{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)};
    const float r0 = v.x - bf_lo(h[0]), r1 = v.y - bf_hi(h[0]), r2 = v.z - bf_lo(h[1]), r3 = v.w - bf_hi(h[1]);
    m = u32x2{pk_bf16(r0, r1), pk_bf16(r2, r3)};
    const float s0 = r0 - bf_lo(m[0]), s1 = r1 - bf_hi(m[0]), s2 = r2 - bf_lo(m[1]), s3 = r3 - bf_hi(m[1]);
    l = u32x2{pk_bf16(s0, s1), pk_bf16(s2, s3)};
}

// tap (dz, dy, dx) of half `hi` of pair p; false: the empty half of the last pair.
//   p =  0 .. 49: (dz, dz + 1) for dz = 0, 2:  p = 25 (dz / 2) + 5 dx + dy
//   p = 50 .. 59: plane dz = 4, (dy, dy + 1) for dy = 0, 2:  p = 50 + 2 dx + dy / 2
//   p = 60, 61  : row (4, 4), (dx, dx + 1) for dx = 0, 2  (round 6; before, every (4, 4, dx) was a pair with an empty half: 65 pairs,
//                 and the wave that got 17 of them set the pace of every chunk);  p = 62: (4, 4, 4) alone
__host__ __device__ __forceinline__ bool x3_pair_tap(int p, int hi, int& dz, int& dy, int& dx) {
    if (p < 50) { const int zp = p / 25, r = p - zp * 25; dx = r / 5; dy = r - dx * 5; dz = 2 * zp + hi; return true; }
    dz = 4;
    if (p < 60) { const int r = p - 50; dx = r / 2; dy = 2 * (r - dx * 2) + hi; return true; }
    dy = 4; dx = 2 * (p - 60) + hi;
    return !(p == 62 && hi);
}

// the inverse: pair and half of tap (dz, dy, dx)
__host__ __device__ __forceinline__ void x3_tap_pair(int dz, int dy, int dx, int& p, int& hi) {
    if (dz < 4) { p = 25 * (dz >> 1) + 5 * dx + dy; hi = dz & 1; }
    else if (dy < 4) { p = 50 + 2 * dx + (dy >> 1); hi = dy & 1; }
    else { p = 60 + (dx >> 1); hi = dx & 1; }
}

// one 16-byte unit (8 consecutive k of one n) of the three filter images:
//   image [k chunk 16][pair 63][n block 16][piece 3][64 lanes][8 k];  lane = (n % 16) + 16 * (k half + 2 * pair half)
//   FWD: k = ci, n = co;  BWD: k = co, n = ci at the flipped tap  (the backward-data convolution)
__device__ __forceinline__ void x3_pack_unit(bool bwd, const float* __restrict__ w, u32x4* __restrict__ out, int I, int O, int ncob, uint32_t u) {
    const uint32_t lane = u & 63;
    uint32_t q = u >> 6;
    const uint32_t qfull = q;
    const uint32_t cob = q % (uint32_t)ncob; q /= (uint32_t)ncob;
    const int p = (int)(q % X3_NPAIR), chunk = (int)(q / X3_NPAIR);
    const int nl = lane & 15, g = lane >> 4, half = g & 1, hi = g >> 1;
    int dz, dy, dx;
    const bool valid = x3_pair_tap(p, hi, dz, dy, dx);
    const int tap = (dz * 5 + dy) * 5 + dx;
    const int k0 = chunk * 16 + half * 8, n = (int)cob * 16 + nl;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (valid) {
        if (!bwd) {
            if (n < O) {
                const float* src = w + ((size_t)tap * I + k0) * O + n;
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k0 + e < I) v[e] = src[(size_t)e * O];
            }
        } else if (n < I) {
            const float* src = w + ((size_t)(124 - tap) * I + n) * O + k0;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (k0 + e < O) v[e] = src[e];
        }
    }
    u32x2 h0, m0, l0, h1, m1, l1;
    x3_split4(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
    x3_split4(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
    u32x4* dst = out + (size_t)qfull * 3 * 64 + lane;
    dst[0] = u32x4{h0[0], h0[1], h1[0], h1[1]};
    dst[64] = u32x4{m0[0], m0[1], m1[0], m1[1]};
    dst[128] = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

// store the three pieces of one unit: v = 8 consecutive k of one n; q = (chunk * X3_NPAIR + pair) * ncob + cob; lane as above
__device__ __forceinline__ void x3_store_unit(u32x4* __restrict__ out, size_t q, uint32_t lane, const float (&v)[8]) {
    u32x2 h0, m0, l0, h1, m1, l1;
    x3_split4(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
    x3_split4(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
    u32x4* dst = out + q * 3 * 64 + lane;
    dst[0] = u32x4{h0[0], h0[1], h1[0], h1[1]};
    dst[64] = u32x4{m0[0], m0[1], m1[0], m1[1]};
    dst[128] = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

__global__ void __launch_bounds__(256) x3_pack_kernel(int bwd, const float* __restrict__ w, u32x4* __restrict__ out, int I, int O, int ncob, uint32_t units) {
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += gridDim.x * blockDim.x) x3_pack_unit(bwd != 0, w, out, I, O, ncob, u);
}

inline void x3_packed_dims(bool bwd, int I, int O, int* nchunk, int* ncob) {
    const int K = bwd ? O : I, N = bwd ? I : O;
    *nchunk = round_up(K, 16) / 16; *ncob = round_up(N, 16) / 16;
}


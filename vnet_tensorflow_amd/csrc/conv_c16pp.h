// conv_c16pp.h -- 16-output-channel 5^3 convolution on bf16 tensors with the FILTER OUT OF LDS and TWO workgroups per CU (round 5,
// VERDICT r4 #3; included by conv_kernels.h in front of the kernel choice conv_fwd_bf16_go).  Replaces tf.nn.convolution
// (layers2.py:59-63 from networks.py:316,333) and its Conv3DBackpropInput (model.py:660) for the full-resolution layers of the
// bf16-storage mode, like conv5_bf16_c16_kernel above.
//
// Why.  conv5_bf16_c16_kernel keeps tile (61 KB) + filter chunk (65 KB) in LDS: one 8-wave workgroup per CU, whose eight waves enter
// the step's non-MFMA phases (prefetch issue, stores, barrier + commit + barrier: ~6 K of 16 K cycles) together; and its MFMA phase
// itself is LDS-BANDWIDTH bound: 13 fragment reads (8 rows + 5 filter) per 20 MFMAs = 0.65 KB per MFMA, x 4 SIMDs at one MFMA per
// 16 cycles = 166 B/clk against the LDS's 128 (measured: 19 cycles per MFMA in the phase; a first ping-pong form of this file that kept
// the filter in LDS -- profiles/probes/conv_c16pp_v1_pingpong.h.txt -- ran at 34: a single wave per SIMD issues its reads and its
// MFMAs in order, so the two queue up instead of overlapping).  Here:
//   * the filter fragments go L2 -> VGPR (the deep-level / f32x3 kernels' way; 65 KB per wave and chunk, ~31 B/clk/CU), two groups ahead;
//   * LDS holds only the tile, so TWO 4-wave workgroups share a CU (2 x 62 KB) and de-phase on their own: one multiplies while the
//     other stores its brick and stages its next tile (the recipe of the fp32 kernel, whose matrix pipe is 99 % busy);
//   * a wave owns a whole z-plane of the 4 x 8 x 16 brick = 8 rows: the 12 row fragments of a (dz pair, dx) feed 40 MFMAs in "j order"
//     (row j serves the output rows m = j - dy): 0.3 KB of LDS per MFMA, reads interleaved with the MFMAs they feed.
// Same MFMA (v_mfma_f32_16x16x32_bf16, K = a pair of taps x 16 cin), same tap-pair order and, per accumulator, the same order of
// additions as conv5_bf16_c16_kernel: the convolution results are bit-identical to it.
#pragma once

namespace {

template <int K, int N, typename F>
__device__ __forceinline__ void pp_for_impl(F&& f) {
    if constexpr (K < N) { f(std::integral_constant<int, K>{}); pp_for_impl<K + 1, N>(f); }
}
template <int N, typename F>
__device__ __forceinline__ void pp_for(F&& f) { pp_for_impl<0, N>(f); }

#ifndef PP_AHEAD
#define PP_AHEAD 3
#endif
constexpr size_t PP_LDS = (size_t)Bf16Geom<4, 8, 16>::TILE_BYTES + 64 * 16 + 4 * 32 * 4;      // tile + dump + statistics scratch

template <bool STATS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_bf16_c16pp_kernel(ConvArgs a) {
    constexpr int TZ = 4, TY = 8, TX = 16, NT = 256;
    using G = Bf16Geom<TZ, TY, TX>;
    using XH = XTileH<G::IZ, G::IY, G::IX, NT>;
    constexpr int ROWB = G::IX * 16, PLANEB = G::IY * G::IX * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* dump = smem + G::TILE_BYTES + lane * 16;
    float* red = reinterpret_cast<float*>(smem + G::TILE_BYTES + 64 * 16);                      // [4 waves][2 x 16]
    const int j = lane & 15, g = lane >> 4, half = g & 1, hi = g >> 1;
    const int vz = wave;                                                   // the wave's z-plane of the brick; rows 0..7

    const int base0 = half * G::PLANE + ((vz * G::IY) * G::IX + j) * 16;
    const unsigned char* bZ = tile + base0 + hi * PLANEB;                // taps (dz, dz+1): lanes 32-63 one tile plane further
    const unsigned char* bY = tile + base0 + hi * ROWB;                  // taps (4, dy), (4, dy+1): one tile row further
    const unsigned char* b0 = tile + base0;                              // single tap (4, 4)

    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    const int G8 = gridDim.x >> 3;                                       // workgroups per XCD (grid is a multiple of 8)
    const int per_xcd = (nbrick + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b_lo = xcd * per_xcd, b_hi = min(nbrick, b_lo + per_xcd);
    if (b_lo + slot >= b_hi) return;
    const int nmine = (b_hi - b_lo - slot + G8 - 1) / G8;                // bricks b_lo + slot + i * G8
    const int nch = a.nchunks;
    const int nsteps = nmine * nch;

    auto brick_origin = [&](int i, int& b, int& bz, int& by, int& bx) {
        int brick = b_lo + slot + i * G8;
        bx = brick % a.nbx; brick /= a.nbx;
        by = brick % a.nby; brick /= a.nby;
        bz = brick % a.nbz; b = brick / a.nbz;
    };

    // ---- filter fragments: generic packed image (ncob = 1), unit (16 B) index ((chunk * 125 + tap) * 2 + cin half) * 32 + cout ----
    // lane l of a fragment: cout l & 15, cin half (l >> 4) & 1, tap of the pair l >> 5: (dz, dz + 1) -> + 25 taps; dz = 4: (dy, dy + 1) -> + 5
    typedef const __attribute__((address_space(1))) u32x4* gw_t;
    const int wco = (half * 32 + j);
    const unsigned laneZ = (unsigned)(hi * 25 * 64 + wco), laneY = (unsigned)(hi * 5 * 64 + wco);
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp);
    auto a_load = [&](auto gic, const u32x4* wchunk, u32x4 (&f)[5]) {
        constexpr int gi = decltype(gic)::value;
        if constexpr (gi < 10) {
            constexpr int zp = gi / 5, dx = gi % 5;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) f[dy] = *(gw_t)(wchunk + (laneZ + (unsigned)(((2 * zp * 5 + dy) * 5 + dx) * 64)));
        } else if constexpr (gi < 15) {
            constexpr int dx = gi - 10;
#pragma unroll
            for (int q = 0; q < 2; ++q) f[q] = *(gw_t)(wchunk + (laneY + (unsigned)(((20 + 2 * q) * 5 + dx) * 64)));
            // the single tap (4, 4, dx): the pair's second tap does not exist -- those lanes read a zero line (address select)
            gw_t s4 = hi ? (gw_t)vnet_zero_line : (gw_t)(wchunk + ((unsigned)wco + (unsigned)((24 * 5 + dx) * 64)));
            f[2] = *s4;
        }
    };

    u32x4 hv[XH::PER];
    auto tile_issue = [&](int i, int ch) {
        int b, bz, by, bx;
        brick_origin(i, b, bz, by, bx);
        XH::template issue_part<0, XH::PER>(hv, reinterpret_cast<const unsigned short*>(a.x0), reinterpret_cast<const unsigned short*>(a.x1),
                                            a.C0, a.C1, ch, b, bz * TZ - 2, by * TY - 2, bx * TX - 2, a.Di, a.Hi, a.Wi, tid);
    };
    auto tile_commit = [&]() { bf16_tile_commit_h<G, XH, 0, XH::PER>(tile, dump, hv, tid); };

    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && 4 * g < a.Cout) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bias4[k] = a.bias[4 * g + k];
    }
    f32x4 acc[8];

    // ---- one (brick, chunk) step's 520 MFMAs: 10 groups (dz pair, dx) of 40 + 5 groups (dz = 4, dx) of 24 ----
    auto mfma_phase = [&](int ch) {
        const u32x4* wchunk = wg + (size_t)ch * 125 * 64;
        u32x4 A[3][5];
        a_load(std::integral_constant<int, 0>{}, wchunk, A[0]);
        a_load(std::integral_constant<int, 1>{}, wchunk, A[1]);
        pp_for<15>([&](auto gic) {
            constexpr int gi = decltype(gic)::value;
            if constexpr (gi + 2 < 15) a_load(std::integral_constant<int, gi + 2>{}, wchunk, A[(gi + 2) % 3]);
            const u32x4 (&Af)[5] = A[gi % 3];
            if constexpr (gi < 10) {
                constexpr int zp = gi / 5, dx = gi % 5;
                const unsigned char* rp = bZ + ((2 * zp * G::IY) * G::IX + dx) * 16;
                bf16x8 R[12];
#pragma unroll
                for (int r = 0; r < PP_AHEAD; ++r) R[r] = *reinterpret_cast<const bf16x8*>(rp + r * ROWB);
                pp_for<12>([&](auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    if constexpr (jj + PP_AHEAD < 12) R[jj + PP_AHEAD] = *reinterpret_cast<const bf16x8*>(rp + (jj + PP_AHEAD) * ROWB);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy) {
                        const int m = jj - dy;
                        if (m >= 0 && m < 8) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Af[dy]), R[jj], acc[m], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                constexpr int dx = gi - 10;
                const unsigned char* pp = bY + ((4 * G::IY) * G::IX + dx) * 16;
                const unsigned char* sp = b0 + ((4 * G::IY + 4) * G::IX + dx) * 16;
                bf16x8 P[10], S[8];
#pragma unroll
                for (int r = 0; r < 10; ++r) P[r] = *reinterpret_cast<const bf16x8*>(pp + r * ROWB);
#pragma unroll
                for (int r = 0; r < 8; ++r) S[r] = *reinterpret_cast<const bf16x8*>(sp + r * ROWB);
                __builtin_amdgcn_sched_barrier(0);
                // sweeps over the eight accumulators: no MFMA waits for its predecessor (per accumulator the order of additions stays)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Af[0]), P[m], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Af[1]), P[m + 2], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Af[2]), S[m], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- epilogue of brick i: lane holds cout 4g..4g+3 of voxel (vz, row m, x = j); bf16 out, optional accumulate / statistics ----
    auto epilogue = [&](int i) {
        int b, bz, by, bx;
        brick_origin(i, b, bz, by, bx);
        const int oz = bz * TZ + vz, ox = bx * TX + j, co = 4 * g;
        const bool colok = oz < a.Do && ox < a.Wo && co < a.Cout;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {                                 // two batches of four rows (a batch keeps its loads in flight together)
            size_t ovs[4]; int cos[4]; bool oks[4]; float e[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int oy = by * TY + h4 * 4 + m;
                oks[m] = colok && oy < a.Ho;
                ovs[m] = oks[m] ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
                cos[m] = oks[m] ? co : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) e[m][q] = acc[h4 * 4 + m][q] + bias4[q];
            }
            epilogue_b16_batch<STATS, 4>(a, ovs, cos, oks, e);
            if constexpr (STATS) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (oks[m]) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { s1[q] += e[m][q]; s2[q] += e[m][q] * e[m][q]; }
                    }
            }
        }
        if constexpr (STATS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1[q] = row16_sum(s1[q]); s2[q] = row16_sum(s2[q]);
                if (j == 0) { red[wave * 32 + co + q] = s1[q]; red[wave * 32 + 16 + co + q] = s2[q]; }
            }
        }
    };

    tile_issue(0, 0);
    tile_commit();
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int i = step / nch, ch = step - i * nch;
        const bool more = step + 1 < nsteps, last = ch == nch - 1;
        if (ch == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#ifndef PP_NO_MFMA
        mfma_phase(ch);
#endif
        __syncthreads();                                   // every wave is done reading the tile
        const int ni = (step + 1) / nch, nc = (step + 1) - ni * nch;
#ifndef PP_NO_TILE
        if (more) { tile_issue(ni, nc); __builtin_amdgcn_sched_barrier(0); }
#endif
#ifndef PP_NO_EPI
        if (last) epilogue(i);
#endif
#ifndef PP_NO_TILE
        if (more) tile_commit();
#endif
        __syncthreads();
        if constexpr (STATS) if (last) stats_row_write<4, 16>(red, a.stats, (size_t)(b_lo + slot + i * G8), 0, a.Cout, tid);
    }
}

}  // namespace

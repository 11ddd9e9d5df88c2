// conv_x3.h -- "f32x3": fp32 5x5x5 convolutions on the bf16 matrix pipe (gfx950), round 5.
//
// tf.nn.convolution (reference layers2.py:59-63, called from networks.py:316,333,346) and its Conv3DBackpropInput (autodiff,
// model.py:660) in fp32 accuracy WITHOUT the fp32 MFMA (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate on this part).
// Every fp32 operand is split exactly into three bf16 pieces
//     x = h + m + l,   h = RNE_bf16(x), m = RNE_bf16(x - h), l = RNE_bf16(x - h - m)      (8 + 8 + 8 significant bits: no remainder)
// and a product x * w is formed from SIX bf16 products accumulated in fp32 by v_mfma_f32_16x16x32_bf16:
//     xh wh + xh wm + xm wh + xm wm + xh wl + xl wh            (the three dropped terms are <= 2^-24 |x w|: below fp32's own rounding)
// measured (profiles/r03_split_bf16_probe.txt): rel-L2 9.9e-7 against float64, the fp32 MFMA itself 1.13e-6.
//
// Kernel (forward and, with the flipped / transposed filter image, backward-data):
//   * D[16 cout][16 voxels] per MFMA, K = 32 = a PAIR of taps x 16 cin (pairs as in the bf16 16-cout kernel: (dz, dz+1) for
//     dz = 0, 2 -> a tile-plane offset for lanes 32..63; dz = 4: (dy, dy+1) -> a tile-row offset; row (4,4): (dx, dx+1) -> a tile-column
//     offset, (4,4,4) alone) = 63 pairs per chunk;
//   * brick 2 x 8 x 16 output voxels x 16 cout per item, persistent 8-wave workgroups (one per CU) walk (brick, cout block) items;
//     the brick + halo of one 16-cin chunk is split while it is committed and sits in LDS as three bf16 images
//     [piece][cin half][voxel][8 cin] (138 KB): a B fragment is one ds_read_b128 per piece, lane base + compile-time offset;
//   * the two z-planes of the brick go to two groups of four waves; the four waves of a group hold the SAME 8 output rows (32
//     accumulator registers) and split K: wave kw owns the tap column dx = kw (12 pairs), a quarter of the column dx = 4 and
//     one of the three pairs of the row (4,4) (16 / 16 / 15 / 16 pairs, rotated by chunk; round 5 had 65 pairs, 16 / 16 / 16 / 17, and the
//     wave with 17 set the pace of every chunk).  So every filter fragment is fetched by exactly two waves of the workgroup,
//     straight from L2 into VGPRs (3 KB per pair, ~16 B/clk/CU) and never touches LDS;
//   * y-sliding reuse: a wave walks the tile rows j of a (dz pair, dx) once and feeds row j to the output rows m = j - dy of
//     every dy it owns: 12 row fragments (x3 pieces) for 40 (row, dy) steps = 240 MFMAs: 0.15 KB of LDS reads per MFMA;
//   * the four partial bricks of a group meet in LDS (tile space, 64 KB) at the end of an item; each wave sums and stores 2 rows;
//   * the next tile (next chunk / next item) is prefetched global -> registers under the MFMAs and split + committed between
//     two barriers.
#pragma once
#include "conv_kernels.h"

namespace {

constexpr int X3_TZ = 2, X3_TY = 8, X3_TX = 16;
constexpr int X3_IZ = X3_TZ + 4, X3_IY = X3_TY + 4, X3_IX = X3_TX + 4;
constexpr int X3_NV = X3_IZ * X3_IY * X3_IX;             // 1440 tile voxels

// Tile geometry of the convolution kernel.
//   W8 = false: brick 2 x 8 x 16, tile 6 x 12 x 20 voxels; the 16 voxels of an MFMA column block are 16 consecutive x.
//   W8 = true (round 6: volumes exactly 8 voxels wide -- the 8^3 level): brick 4 x 8 x 8.  The 16 voxels of a column block are
//     8 x of plane z and the same 8 x of plane z + 2: lanes 8..15 read the tile two planes further.  Everything else -- tap pairs,
//     K split over the waves, y-sliding reuse, the reduction -- is the same code on other constants.  Tile 8 x 12 x 12 voxels; the plane
//     pitch carries 64 bytes of padding so that the two 8-lane halves of a ds_read_b128 service group (lanes 0-3 | 12-15 | 20-27) sit
//     128 bytes apart modulo 256: conflict-free like the contiguous 16-lane row of the wide brick.
template <bool W8>
struct X3G {
    static constexpr int TZ = W8 ? 4 : 2;
    static constexpr int IZ = TZ + 4, IY = X3_IY, IX = W8 ? 12 : X3_IX;
    static constexpr int ROWB = IX * 16;
    static constexpr int ZB = IY * ROWB + (W8 ? 64 : 0);
    static constexpr int PLANEB = IZ * ZB;               // one (piece, cin half) plane: 23040 B / 18944 B (multiples of 256: bank-neutral)
    static constexpr int PIECEB = 2 * PLANEB;
    static constexpr int TILEB = 3 * PIECEB;             // 138240 / 113664
    static constexpr int LDS = TILEB + 2 * 8 * 32 * 4;   // + epilogue statistics [NB <= 2][8 waves][32]
    static constexpr int SPER = W8 ? 8 : 12;             // staging loads per thread and tile
};
static_assert(X3G<false>::PLANEB == X3_NV * 16 && X3G<false>::PLANEB % 256 == 0 && X3G<true>::PLANEB % 256 == 0, "tile planes");
static_assert((2 * X3G<true>::ZB) % 256 == 128, "lanes 8..15 of the narrow brick: the other half of the 256-byte bank row");
constexpr int X3_LDS = X3G<false>::LDS;

// Compile-time loop: body(std::integral_constant<int, k>) for k = 0 .. N - 1.  NOT `#pragma unroll`: when hipcc's IndVarSimplify visits
// a not-yet-unrolled loop it sinks every side-effect-free instruction of the loop's preheader that the loop does not use to behind the
// loop (sinkUnusedInvariants) -- i.e. the tails of the MFMA chains of the piece in front of it, whose B fragments then stay live
// across the whole piece: +88 VGPRs, 49 spilled, scratch reloads between the prefetch loads (found by -opt-bisect-limit, round 5).
template <int K, int N, typename F>
__device__ __forceinline__ void x3_for_impl(F&& f) {
    if constexpr (K < N) { f(std::integral_constant<int, K>{}); x3_for_impl<K + 1, N>(f); }
}
template <int N, typename F>
__device__ __forceinline__ void x3_for(F&& f) { x3_for_impl<0, N>(f); }

// ---- the K loop pieces -------------------------------------------------------------------------------------------------------
// A fragments of NP consecutive pairs (piece-major per pair: h, m, l), global -> registers
// (a UNIFORM base + a uniform 32-bit unit offset + the lane: the saddr form of global_load; with the lane folded into a 64-bit vector
//  pointer every fragment address is a 64-bit vector multiply-add and a register pair)
typedef const __attribute__((address_space(1))) u32x4* x3_gw_t;
template <int NP>
__device__ __forceinline__ void x3_load_a(bf16x8 (&A)[NP][3], const u32x4* __restrict__ wbase, unsigned u0, unsigned ustride, int lane) {
    x3_for<NP * 3>([&](auto I) {
        constexpr int p = decltype(I)::value / 3, s = decltype(I)::value % 3;
        A[p][s] = __builtin_bit_cast(bf16x8, *(x3_gw_t)(wbase + (u0 + (unsigned)p * ustride + (unsigned)s * 64u) + lane));
    });
}
__device__ __forceinline__ void x3_load_a1(bf16x8 (&A)[3], const u32x4* __restrict__ wbase, unsigned u0, int lane) {
    x3_for<3>([&](auto I) {
        constexpr int s = decltype(I)::value;
        A[s] = __builtin_bit_cast(bf16x8, *(x3_gw_t)(wbase + (u0 + (unsigned)s * 64u) + lane));
    });
}

template <typename G>
__device__ __forceinline__ void x3_read_b(bf16x8 (&B)[3], const unsigned char* p) {
    B[0] = *reinterpret_cast<const bf16x8*>(p);
    B[1] = *reinterpret_cast<const bf16x8*>(p + G::PIECEB);
    B[2] = *reinterpret_cast<const bf16x8*>(p + 2 * G::PIECEB);
}

// the six products of one (A pair, B row): piece indices (A, B); small terms first
#define X3_PROD_A(k) ((k) == 0 ? 2 : (k) == 1 ? 0 : (k) == 2 ? 1 : (k) == 3 ? 1 : (k) == 4 ? 0 : 0)
#define X3_PROD_B(k) ((k) == 0 ? 0 : (k) == 1 ? 2 : (k) == 2 ? 1 : (k) == 3 ? 0 : (k) == 4 ? 1 : 0)

// taps (dz pair, dy = DY0 .. DY0 + NDY - 1, dx) on the 8 output rows of this wave's plane: tile rows jr = DY0 .. DY0 + NDY + 6, row jr
// feeds output row m = jr - dy.  bz: lane base + plane pair + dx.
// (A0: index of the first of the NDY pairs in A -- slices of a larger register array without a cast, which would send it to scratch)
template <typename G, int DY0, int NDY, int A0 = 0, int NA = NDY>
__device__ __forceinline__ void x3_zrows(f32x4 (&acc)[8], const unsigned char* bz, const bf16x8 (&A)[NA][3]) {
    constexpr int J0 = DY0, J1 = DY0 + NDY + 7;
    bf16x8 Bn[3];
    x3_read_b<G>(Bn, bz + J0 * G::ROWB);
    x3_for<J1 - J0>([&](auto JI) {
        constexpr int jr = J0 + decltype(JI)::value;
        bf16x8 B[3] = {Bn[0], Bn[1], Bn[2]};
        if constexpr (jr + 1 < J1) x3_read_b<G>(Bn, bz + (jr + 1) * G::ROWB);
        __builtin_amdgcn_sched_barrier(0);
        x3_for<6 * NDY>([&](auto KI) {
            constexpr int k = decltype(KI)::value / NDY, d = decltype(KI)::value % NDY, m = jr - DY0 - d;
            if constexpr (m >= 0 && m < 8)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[A0 + d][X3_PROD_A(k)], B[X3_PROD_B(k)], acc[m], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// plane dz = 4 of column dx: pairs (dy 0|1), (dy 2|3) from row-pair fragments (lanes 32..63 one row further): A[A0], A[A0 + 1]
template <typename G, int A0, int NA>
__device__ __forceinline__ void x3_yrows_pairs(f32x4 (&acc)[8], const unsigned char* by, const bf16x8 (&A)[NA][3]) {
    bf16x8 Bn[3];
    x3_read_b<G>(Bn, by);
    x3_for<10>([&](auto KI) {                       // row pair (k | k + 1): output row k with q = 0, row k - 2 with q = 1
        constexpr int k = decltype(KI)::value;
        bf16x8 B[3] = {Bn[0], Bn[1], Bn[2]};
        if constexpr (k + 1 < 10) x3_read_b<G>(Bn, by + (k + 1) * G::ROWB);
        __builtin_amdgcn_sched_barrier(0);
        x3_for<6>([&](auto TI) {
            constexpr int t = decltype(TI)::value;
            if constexpr (k < 8) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[A0][X3_PROD_A(t)], B[X3_PROD_B(t)], acc[k], 0, 0, 0);
            if constexpr (k >= 2) acc[k - 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[A0 + 1][X3_PROD_A(t)], B[X3_PROD_B(t)], acc[k - 2], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}
// row (4, 4): single rows m + 4.  Called with the base b0: tap (4, 4, dx) alone (the filter's second half is zero, both lane halves read
// the same column); with bX: the taps (4, 4, dx), (4, 4, dx + 1) (lanes 32..63 one column further): A[A0]
template <typename G, int A0, int NA>
__device__ __forceinline__ void x3_yrows_single(f32x4 (&acc)[8], const unsigned char* b0, const bf16x8 (&A)[NA][3]) {
    bf16x8 Bn[3];
    x3_read_b<G>(Bn, b0 + 4 * G::ROWB);
    x3_for<8>([&](auto MI) {
        constexpr int m = decltype(MI)::value;
        bf16x8 B[3] = {Bn[0], Bn[1], Bn[2]};
        if constexpr (m + 1 < 8) x3_read_b<G>(Bn, b0 + (m + 5) * G::ROWB);
        __builtin_amdgcn_sched_barrier(0);
        x3_for<6>([&](auto TI) {
            constexpr int t = decltype(TI)::value;
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[A0][X3_PROD_A(t)], B[X3_PROD_B(t)], acc[m], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// NB: 16-cout blocks per item (1 or 2).  With two, a step runs the four pieces twice on the SAME tile -- once per block, one block's
// accumulators and filter fragments at a time -- so the tile commit, its two barriers and the item's reduction are paid once per
// 2 x 6048 MFMAs instead of once per 6048 (63 pairs x 8 rows x 6 products x 2 planes).
template <bool STATS, int NB, bool W8 = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_x3_kernel(ConvArgs a) {
    constexpr int NT = 512;
    using G = X3G<W8>;
    using XT = XTile<X3_IZ, X3_IY, X3_IX, NT>;              // (staging shape of the wide brick)
    static_assert(XT::PER * XT::RPI == XT::ROWS && XT::PER == X3G<false>::SPER, "every staging pass covers whole tile rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    float* red = reinterpret_cast<float*>(smem);                              // after an item's K loop: [8 waves][NB][8 rows][64 lanes][4]
    float* sred = reinterpret_cast<float*>(smem + G::TILEB);                  // [8 waves][2 x 16] epilogue statistics
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4, half = g & 1, hi = g >> 1;
    const int grp = wave >> 2, kw = wave & 3;

    // (narrow brick: lanes 8..15 of a column block are the plane two further)
    const int base0 = W8 ? half * G::PLANEB + (grp + 2 * (j >> 3)) * G::ZB + (j & 7) * 16
                         : half * G::PLANEB + ((grp * G::IY) * G::IX + j) * 16;
    const unsigned char* bZ = tile + base0 + hi * G::ZB;                      // taps (dz, dz + 1)
    const unsigned char* bY = tile + base0 + hi * G::ROWB + 4 * G::ZB;        // taps (4, dy), (4, dy + 1)
    const unsigned char* b0 = tile + base0 + 4 * G::ZB;                       // tap (4, 4, dx) alone
    const unsigned char* bX = b0 + hi * 16;                                   // taps (4, 4, dx), (4, 4, dx + 1)

    const int ncob = a.CoutP >> 4, ncobg = ncob / NB;                         // 16-cout blocks; groups of NB blocks = items per brick
    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    const int nks = a.nz;                                                     // K splits (chunk ranges of a.cps chunks; partial slabs in a.part)
    const int nitems = nbrick * ncobg * nks;
    const int G8 = gridDim.x >> 3;                                            // workgroups per XCD (grid is a multiple of 8)
    const int per_xcd = (nitems + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int i_lo = xcd * per_xcd, i_hi = min(nitems, i_lo + per_xcd);
    if (i_lo + slot >= i_hi) return;
    const int nmine = (i_hi - i_lo - slot + G8 - 1) / G8;                     // items i_lo + slot + i * G8
    const int nch = a.cps;                                                    // chunks per item
    const int nsteps = nmine * nch;
    const unsigned astride = (unsigned)ncob * 3u * 64u;                       // u32x4 units between consecutive pairs
    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wp);

    auto item_coords = [&](int it, int& b, int& bz, int& by, int& bx, int& cob) {
        int item = (i_lo + slot + it * G8) / nks;
        cob = (item % ncobg) * NB; item /= ncobg;
        bx = item % a.nbx; item /= a.nbx;
        by = item % a.nby; item /= a.nby;
        bz = item % a.nbz; b = item / a.nbz;
    };
    // Tile staging: a thread owns one (x, channel quad) column and walks the 72 (z, y) rows of the tile in steps of 6: row = r0 + 6 k,
    // i.e. iz = k >> 1, iy = r0 + 6 (k & 1).  Loads are 16 bytes from ONE uniform 64-bit base (source tensor, sample, chunk) + a 32-bit
    // element offset; out-of-volume units load a device zero line -- a select on the ADDRESS: a select on the data makes hipcc wait
    // for the whole tile right where it was issued (s_memtime stamps: 4 K cycles in front of every step's first MFMA).
    typedef const __attribute__((address_space(1))) f32x4* gf4_t;
    f32x4 pv[G::SPER];
    // (480 threads cover the 6 x 80 columns of a pass; the last 32 repeat the work of the 32 threads one row block before them -- same
    //  loads, same stores -- instead of being masked: a per-row select on the store address is loop-invariant, and hipcc hoists all 36
    //  of them out of the step loop and spills them)
    // Narrow brick: only the 8 x 4 (x, channel quad) columns inside the volume are staged (the x halo of a volume exactly 8 wide is
    // zero for every brick: written once per item, zero_halo below); a pass is one whole tile plane = 12 rows x 32 columns = 384 threads,
    // the last 128 repeat the 128 in front of them; pass k is plane k.
    const int stid = W8 ? (tid < 384 ? tid : tid - 128) : (tid < XT::RPI * XT::COLS ? tid : tid - XT::COLS);
    constexpr int SCOLS = W8 ? 32 : XT::COLS;
    const int sr0 = stid / SCOLS, scol = stid - sr0 * SCOLS;
    const int six = scol >> 2, scq = scol & 3;
    auto tile_issue = [&](int step) {
        const int it = step / nch, ch = ((i_lo + slot + it * G8) % nks) * nch + (step - it * nch);
        int b, bz, by, bx, cob;
        item_coords(it, b, bz, by, bx, cob);
        const int c0 = ch * 16;
        const bool first = c0 < a.C0;
        const int cs = first ? a.C0 : a.C1;
        const float* src = (first ? a.x0 + c0 : a.x1 + (c0 - a.C0)) + (size_t)b * a.Di * a.Hi * a.Wi * cs;
        if constexpr (W8) {
            const int gz0 = bz * G::TZ - 2, gy = by * X3_TY - 2 + sr0;
            const bool yok = (unsigned)gy < (unsigned)a.Hi;
            const int rs = a.Wi * cs;
            const int off0 = (gy * a.Wi + six) * cs + scq * 4;
            x3_for<G::SPER>([&](auto KI) {
                constexpr int k = decltype(KI)::value;
                const int gz = gz0 + k;
                const bool ok = yok && (unsigned)gz < (unsigned)a.Di;
                const int off = off0 + gz * a.Hi * rs;
                gf4_t p = ok ? (gf4_t)(src + off) : (gf4_t)(vnet_zero_line);
                pv[k] = *p;
            });
        } else {
            const int gz0 = bz * X3_TZ - 2, gy = by * X3_TY - 2 + sr0, gx = bx * X3_TX - 2 + six;
            const bool xok = (unsigned)gx < (unsigned)a.Wi;
            const bool yok0 = xok && (unsigned)gy < (unsigned)a.Hi, yok1 = xok && (unsigned)(gy + 6) < (unsigned)a.Hi;
            const int rs = a.Wi * cs;
            const int off0 = (gy * a.Wi + gx) * cs + scq * 4;
            x3_for<XT::PER>([&](auto KI) {
                constexpr int k = decltype(KI)::value;
                const int gz = gz0 + (k >> 1);
                const bool ok = ((k & 1) ? yok1 : yok0) && (unsigned)gz < (unsigned)a.Di;
                const int off = off0 + gz * a.Hi * rs + (k & 1) * 6 * rs;
                gf4_t p = ok ? (gf4_t)(src + off) : (gf4_t)(vnet_zero_line);
                pv[k] = *p;
            });
        }
    };
    const unsigned cbase = (unsigned)((scq >> 1) * G::PLANEB + (six + (W8 ? 2 : 0)) * 16 + (scq & 1) * 8 + sr0 * G::ROWB);
    // sgn: 0, or 0x80000000 for the tiles of the odd chunks of an item (SIGN ALTERNATION, see the step loop)
    auto tile_commit = [&](const unsigned sgn) {
        // (opaque per call: the 36 store addresses depend only on the thread, hipcc would otherwise compute them once in front of the
        //  step loop and keep -- spill -- them for the whole kernel: scratch reloads between the prefetch loads)
        unsigned cb = cbase;
        asm volatile("" : "+v"(cb));
        unsigned char* base = tile + cb;
        x3_for<G::SPER>([&](auto KI) {
            constexpr int k = decltype(KI)::value;
            u32x2 h, m, l;
            x3_split4(make_float4(__uint_as_float(__float_as_uint(pv[k][0]) ^ sgn), __uint_as_float(__float_as_uint(pv[k][1]) ^ sgn),
                                  __uint_as_float(__float_as_uint(pv[k][2]) ^ sgn), __uint_as_float(__float_as_uint(pv[k][3]) ^ sgn)), h, m, l);
            unsigned char* dst = base + k * (W8 ? G::ZB : XT::RPI * G::ROWB);
            *reinterpret_cast<u32x2*>(dst) = h;
            *reinterpret_cast<u32x2*>(dst + G::PIECEB) = m;
            *reinterpret_cast<u32x2*>(dst + 2 * G::PIECEB) = l;
        });
    };
    // narrow brick: the four x-halo columns (0, 1, 10, 11) of the 96 rows x 2 cin halves x 3 pieces.  They are never staged, but the
    // item's reduction runs through the head of the tile space: rewritten at the start and after every reduction (addresses disjoint
    // from the commit's, so both share one barrier interval)
    auto zero_halo = [&]() {
        if constexpr (W8) {
            int t0 = tid;
            asm volatile("" : "+v"(t0));
#pragma unroll 1
            for (int q = t0; q < 3 * 2 * 96 * 4; q += NT) {
                const int c = q & 3, row = (q >> 2) % 96, ph = q / (96 * 4);
                const int iz = row / 12, iy = row - iz * 12;
                *reinterpret_cast<u32x4*>(tile + ph * G::PLANEB + iz * G::ZB + iy * G::ROWB + (c < 2 ? c : c + 8) * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    zero_halo();
    tile_issue(0);
    tile_commit(0u);
    __syncthreads();

    f32x4 acc[NB][8];
    VNET_STAMP_DECL;
    for (int step = 0; step < nsteps; ++step) {
        VNET_STAMP_STEP(step);
        VNET_STAMP(0);
        const int it = step / nch, lc = step - it * nch, ks = (i_lo + slot + it * G8) % nks, ch = ks * nch + lc;
        const bool first = lc == 0, last = lc == nch - 1, more = step + 1 < nsteps;
        int b, bz, by, bx, cob;
        item_coords(it, b, bz, by, bx, cob);
        // SIGN ALTERNATION (round 6; the filter gradient below has the measurements): the bf16 instruction's accumulation leaves
        // every result slightly BELOW the exact sum, whatever the operands' signs -- per output -4.8e-8 of rms |y| at K = 2000, invisible
        // next to the 3e-7 of random rounding, but the SAME sign in every voxel, so anything that sums an output over the volume (a
        // batch-norm's d beta, a filter gradient) adds it up coherently.  The tiles of an item's odd chunks are therefore committed
        // NEGATED and the accumulators change sign at every chunk boundary (both exact): consecutive chunks carry the offset with
        // opposite signs.  (Layers with ONE 16-channel chunk -- 16 -> 16 at 128^3 -- have nothing to alternate with and keep theirs.)
        if (first) x3_for<8 * NB>([&](auto MI) { acc[decltype(MI)::value / 8][decltype(MI)::value % 8] = f32x4{0.f, 0.f, 0.f, 0.f}; });
        else x3_for<8 * NB>([&](auto MI) { acc[decltype(MI)::value / 8][decltype(MI)::value % 8] = -acc[decltype(MI)::value / 8][decltype(MI)::value % 8]; });
        const int kc = (kw + ch) & 3;                                       // tap column of this wave in this chunk
        // The two waves of a SIMD (w, w + 4) run the same pieces; the matrix pipe goes to the OLDER one whenever both have an MFMA ready
        // (s_memtime stamps: waves 0-3 finished their 780 MFMAs after 19 K cycles, waves 4-7 -- alone, at a single wave's rate -- after
        // 31 K).  Priority for the younger half during the first two pieces, for the older one afterwards: both halves reach the barrier
        // together, and because the halves are then half a pass apart, the L2 round trip of a piece's filter fragments (loaded at the
        // piece's head, into ONE register set) hides under the other half's MFMAs: a prefetch one piece ahead (a second set of 60
        // registers) measured the same (profiles/r05_x3_experiments.txt) and is what kept two cout blocks per item from fitting.
        x3_for<NB>([&](auto NBI) {
            constexpr int nb = decltype(NBI)::value;
            const unsigned wc = (unsigned)(ch * X3_NPAIR * ncob + cob + nb) * 3u * 64u;      // pair 0 of (chunk, cout block), in 16-byte units
            bf16x8 A[5][3];
            if (grp) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            x3_load_a<5>(A, wbase, wc + (unsigned)(kc * 5) * astride, astride, lane);               // (dz 0|1, dx = kc)
            __builtin_amdgcn_sched_barrier(0);
            x3_zrows<G, 0, 5>(acc[nb], bZ + kc * 16, A);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (nb == 0) VNET_STAMP(1);
            x3_load_a<5>(A, wbase, wc + (unsigned)(25 + kc * 5) * astride, astride, lane);          // (dz 2|3, dx = kc)
            __builtin_amdgcn_sched_barrier(0);
            x3_zrows<G, 0, 5>(acc[nb], bZ + 2 * G::ZB + kc * 16, A);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (nb == 0) VNET_STAMP(2);
            x3_load_a1(A[0], wbase, wc + (unsigned)(50 + kc * 2) * astride, lane);                  // (dz 4, dy 0|1, dx = kc)
            x3_load_a1(A[1], wbase, wc + (unsigned)(51 + kc * 2) * astride, lane);                  // (dz 4, dy 2|3, dx = kc)
            x3_load_a1(A[2], wbase, wc + (unsigned)(60 + (kc & 1)) * astride, lane);                // taps (4, 4, 0|1) with kc == 0, (4, 4, 2|3) with kc == 1
            x3_load_a1(A[3], wbase, wc + 62u * astride, lane);                                     // tap (4, 4, 4): goes with kc == 3
            __builtin_amdgcn_sched_barrier(0);
            if (grp) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
            // the next tile's loads go out in the LAST pass, behind this piece's filter loads (vmcnt counts in order) and late enough
            // that its 48 registers are live for two pieces only; unconditional: a branch around an issue merges two vmcnt states
            if constexpr (nb == NB - 1) tile_issue(min(step + 1, nsteps - 1));
            __builtin_amdgcn_sched_barrier(0);
            x3_yrows_pairs<G, 0, 5>(acc[nb], bY + kc * 16, A);
            if (kc < 2) x3_yrows_single<G, 2, 5>(acc[nb], bX + kc * 32, A);          // columns (0|1) / (2|3) of row (4, 4)
            if (kc == 3) x3_yrows_single<G, 3, 5>(acc[nb], b0 + 4 * 16, A);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (nb == 0) VNET_STAMP(3);
            // the column dx = 4 in four parts (3 / 3 / 3 / 3 + 1 pairs): pairs 20-22 | 23, 24, 45 | 46-48 | 49, 58, 59 (+ 62 above)
            {
                const int p0 = kc == 0 ? 20 : kc == 1 ? 23 : kc == 2 ? 46 : 49;
                const int p1 = kc == 0 ? 21 : kc == 1 ? 24 : kc == 2 ? 47 : 58;
                const int p2 = kc == 0 ? 22 : kc == 1 ? 45 : kc == 2 ? 48 : 59;
                x3_load_a1(A[0], wbase, wc + (unsigned)p0 * astride, lane);
                x3_load_a1(A[1], wbase, wc + (unsigned)p1 * astride, lane);
                x3_load_a1(A[2], wbase, wc + (unsigned)p2 * astride, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kc == 0) {
                x3_zrows<G, 0, 3, 0, 5>(acc[nb], bZ + 4 * 16, A);
            } else if (kc == 1) {
                x3_zrows<G, 3, 2, 0, 5>(acc[nb], bZ + 4 * 16, A);
                x3_zrows<G, 0, 1, 2, 5>(acc[nb], bZ + 2 * G::ZB + 4 * 16, A);
            } else if (kc == 2) {
                x3_zrows<G, 1, 3, 0, 5>(acc[nb], bZ + 2 * G::ZB + 4 * 16, A);
            } else {
                x3_zrows<G, 4, 1, 0, 5>(acc[nb], bZ + 2 * G::ZB + 4 * 16, A);
                x3_yrows_pairs<G, 1, 5>(acc[nb], bY + 4 * 16, A);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        VNET_STAMP(4);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();                                   // every wave is done reading the tile
        VNET_STAMP(5);
        if (last) {
            // the four partial bricks of each group meet in LDS; wave kw sums and stores the output rows 2 kw, 2 kw + 1
            const bool flip = lc & 1;                      // the item's last chunk ran negated
            x3_for<8 * NB>([&](auto MI) {
                constexpr int nb = decltype(MI)::value / 8, m = decltype(MI)::value % 8;
                *reinterpret_cast<f32x4*>(red + (((wave * NB + nb) * 8 + m) * 64 + lane) * 4) = flip ? -acc[nb][m] : acc[nb][m];
            });
            __syncthreads();
            x3_for<NB>([&](auto NBI) {
            constexpr int nb = decltype(NBI)::value;
            const int co0 = (cob + nb) * 16, co = co0 + 4 * g;
            f32x4 o[2];
            x3_for<2>([&](auto TI) {
                constexpr int t = decltype(TI)::value;
                const int m = 2 * kw + t;
                constexpr int WS = NB * 8 * 64 * 4;                    // floats between two waves' partial bricks
                const float* rp = red + (((grp * 4 * NB + nb) * 8 + m) * 64 + lane) * 4;
                o[t] = (*reinterpret_cast<const f32x4*>(rp) + *reinterpret_cast<const f32x4*>(rp + WS)) +
                       (*reinterpret_cast<const f32x4*>(rp + 2 * WS) + *reinterpret_cast<const f32x4*>(rp + 3 * WS));
            });
            float bias4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && co < a.Cout) {
                const float4 bb = *reinterpret_cast<const float4*>(a.bias + co);
                bias4[0] = bb.x; bias4[1] = bb.y; bias4[2] = bb.z; bias4[3] = bb.w;
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            const int oz = W8 ? bz * G::TZ + grp + 2 * (j >> 3) : bz * X3_TZ + grp, ox = W8 ? (j & 7) : bx * X3_TX + j;
            x3_for<2>([&](auto TI) {
                constexpr int t = decltype(TI)::value;
                const int oy = by * X3_TY + 2 * kw + t;
                if (oz < a.Do && oy < a.Ho && ox < a.Wo && co < a.Cout) {
                    const size_t ov = ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
                    if (a.part) {                          // K split: the raw partial sums; bias / accumulate / statistics belong to the reduce
                        *reinterpret_cast<f32x4*>(a.part + (size_t)ks * a.part_stride + ov * a.CoutP + co) = o[t];
                        return;
                    }
                    float e[4] = {o[t][0] + bias4[0], o[t][1] + bias4[1], o[t][2] + bias4[2], o[t][3] + bias4[3]};
                    if constexpr (STATS) {
                        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (a.res) rr = *reinterpret_cast<const float4*>(a.res + ov * a.Cout + co);
                        const float vv[4] = {e[0] + rr.x, e[1] + rr.y, e[2] + rr.z, e[3] + rr.w};
                        s1[0] += vv[0]; s1[1] += vv[1]; s1[2] += vv[2]; s1[3] += vv[3];
                        s2[0] += vv[0] * vv[0]; s2[1] += vv[1] * vv[1]; s2[2] += vv[2] * vv[2]; s2[3] += vv[3] * vv[3];
                    }
                    float* p = (co < a.Cy0) ? a.y0 + ov * a.Cy0 + co : a.y1 + ov * a.Cy1 + (co - a.Cy0);
                    if (a.accum) {
                        const float4 old = *reinterpret_cast<const float4*>(a.accsrc ? a.accsrc + ov * a.Cy0 + co : p);
                        e[0] += old.x; e[1] += old.y; e[2] += old.z; e[3] += old.w;
                    }
                    *reinterpret_cast<float4*>(p) = make_float4(e[0], e[1], e[2], e[3]);
                }
            });
            if constexpr (STATS) {
                x3_for<4>([&](auto KI) {
                    constexpr int k = decltype(KI)::value;
                    s1[k] = row16_sum(s1[k]); s2[k] = row16_sum(s2[k]);
                    if (j == 0) { sred[(nb * 8 + wave) * 32 + 4 * g + k] = s1[k]; sred[(nb * 8 + wave) * 32 + 16 + 4 * g + k] = s2[k]; }
                });
            }
            });
            __syncthreads();                               // the partial bricks are read; the statistics of all waves are in sred
            if constexpr (STATS) {
                int brick = (i_lo + slot + it * G8) / (ncobg * nks);
                x3_for<NB>([&](auto NBI) {
                    constexpr int nb = decltype(NBI)::value;
                    stats_row_write<8, 16>(sred + nb * 8 * 32, a.stats, (size_t)brick, (cob + nb) * 16, a.Cout, tid);
                });
            }
        }
        VNET_STAMP(6);
        if (last) zero_halo();
        if (more) tile_commit(((last ? 0 : lc + 1) & 1) ? 0x80000000u : 0u);
        VNET_STAMP(7);
        __syncthreads();
        VNET_STAMP(8);
        VNET_STAMP_FLUSH(step, wave, lane, 9);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Filter gradient (Conv3DBackpropFilterV2 behind model.py:660) with the same six-product arithmetic:
//   D[16 cout][16 cin] per tap += A[cout][k = 32 voxels] * B[32 voxels][cin, tap-shifted]      (v_mfma_f32_16x16x32_bf16 x 6)
// Structure of the bf16 row-reuse kernel (wgrad5_bf16_rr_kernel), on a 2 x 8 x 16 brick: x tile (6 x 12 x 20 voxels x 16 cin)
// and dy tile (256 voxels x 16 cout) are split while they are committed and sit in LDS as three bf16 images [piece][voxel][16 ch]
// (32-byte rows: 138 + 24 KB -- all of the CU's LDS); fragments by ds_read_b64_tr_b16 (a tap shift is a whole number of rows).
// A k-step is two x-rows (y, y + 1) of one output plane; lane group g: row g & 1, x half g >> 1 (row pitch 640 B = 128 mod 256:
// the two groups of an LDS service half use disjoint banks; the dy tile's 512-byte rows swap their 128-byte halves on odd rows
// for the same reason).  The B fragment of (k-step s, tap dy) is row pair j = 2 s + dy: a wave owns three (dz, dx) columns with all
// five dy, one column at a time, and keeps a sliding window of five row-pair fragments (x3 pieces): 2 new B + 1 A fragment triple per
// 30 MFMAs.  The 25th column (4, 4) gives one tap to each of the waves 0..4.  One (16 cin, 16 cout) block per workgroup, bricks
// split over nsplit workgroups, next brick's tiles prefetched global -> registers under the MFMAs, partial slabs + the usual reduce.
// ------------------------------------------------------------------------------------------------------------------------------
// Tile geometry of the filter-gradient kernel.  W8 (volumes exactly 8 wide, brick 4 x 8 x 8): the k-step's two x halves (lane groups
// 2, 3) are the planes z + 2 instead of the columns x + 8 -- an x tile of 8 x 12 x 12 voxels; in the dy tile a 16-slot row holds the
// 8 voxels of plane z and the 8 of plane z + 2, so only the staging coordinates differ there.
template <bool W8>
struct XWG {
    static constexpr int TZ = W8 ? 4 : 2;
    static constexpr int IZ = TZ + 4, IX = W8 ? 12 : X3_IX;
    static constexpr int XROW = IX * 32, XPLANE = X3_IY * XROW;      // row pitch 640 / 384 B: both 128 mod 256 (see below)
    static constexpr int XPB = IZ * XPLANE;                          // one piece of the x tile: 46080 / 36864 B
    static constexpr int DPB = 2 * X3_TY * 16 * 32;                  // one piece of the dy tile: 8192 B
    static constexpr int LDS = 3 * XPB + 3 * DPB;                    // 162816 (of 163840) / 135168
    static constexpr int XHALF = W8 ? 2 * XPLANE : 8 * 32;           // lane groups 2, 3
    static constexpr int XPER = W8 ? 8 : 12;                         // staging loads per thread and x tile
};
static_assert(XWG<false>::XROW % 256 == 128 && XWG<true>::XROW % 256 == 128, "row pairs of a k-step on disjoint banks");
constexpr int XW_LDS = XWG<false>::LDS;

// three-piece fragment: 2 transpose reads per piece (voxels +0..3 at lo, +4..7 at hi)
__device__ __forceinline__ void xw_frag(bf16x8 (&F)[3], const unsigned char* lo, const unsigned char* hi, int pieceb) {
    typedef s16x4 __attribute__((address_space(3))) * lp;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lo + s * pieceb));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(hi + s * pieceb));
        const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        F[s] = __builtin_bit_cast(bf16x8, v);
    }
}

__device__ __forceinline__ void xw_mfma6(f32x4& acc, const bf16x8 (&A)[3], const bf16x8 (&B)[3]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[X3_PROD_A(k)], B[X3_PROD_B(k)], acc, 0, 0, 0);
}

template <bool W8>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) wgrad5_x3_kernel(WgradArgs a) {
    constexpr int NT = 512;
    using G = XWG<W8>;
    constexpr int XW_XPB = G::XPB, XW_DPB = G::DPB, XW_XROW = G::XROW, XW_XPLANE = G::XPLANE;
    using XT = XTile<X3_IZ, X3_IY, X3_IX, NT>;              // (staging shape of the wide brick)
    static_assert(XT::PER == XWG<false>::XPER, "staging passes of the wide brick");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;
    unsigned char* dyt = smem + 3 * XW_XPB;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int split = blockIdx.x;
    const int chunk = blockIdx.y / a.ncob, cob = blockIdx.y - chunk * a.ncob;
    const int co0 = cob * 16;

    const int lane_x = W8 ? ((g & 1) * G::IX + (i >> 2)) * 32 + (g >> 1) * G::XHALF + (i & 3) * 8
                            : (((g & 1) * X3_IX + (g >> 1) * 8) + (i >> 2)) * 32 + (i & 3) * 8;
    const unsigned char* pb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int q = 3 * wave + c, dz = q / 5, dx = q - dz * 5;
        pb[c] = xt + dz * XW_XPLANE + dx * 32 + lane_x;
    }
    const bool extra = wave < 5;
    const unsigned char* pe = xt + 4 * XW_XPLANE + (extra ? wave : 0) * XW_XROW + 4 * 32 + lane_x;
    const int lane_d = (((g & 1) * 16 + (g >> 1) * 8) + (i >> 2)) * 32 + (i & 3) * 8;
    const unsigned char* pa_lo = dyt + lane_d + (g & 1) * 128;
    const unsigned char* pa_hi = dyt + lane_d + (1 - (g & 1)) * 128;

    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 px[G::XPER], pd[2];
    auto brick_coords = [&](int brick, int& b, int& bz, int& by, int& bx) {
        bx = brick % a.nbx; brick /= a.nbx;
        by = brick % a.nby; brick /= a.nby;
        bz = brick % a.nbz; b = brick / a.nbz;
    };
    auto issue = [&](int brick) {
        int b, bz, by, bx;
        brick_coords(brick, b, bz, by, bx);
        if constexpr (W8) {
            // only the 8 x 4 (x, channel quad) columns inside the volume (its x halo is zero for every brick: written once, below);
            // a pass is one tile plane = 12 rows x 32 columns = 384 threads, the last 128 repeat the 128 in front of them
            const int stid = tid < 384 ? tid : tid - 128;
            const int iy = stid >> 5, col = stid & 31;
            const int c = chunk * 16 + (col & 3) * 4;
            const bool first = c < a.C0;
            const int cs = first ? a.C0 : a.C1;
            const float* src = (first ? a.x0 + c : a.x1 + (c - a.C0)) + (size_t)b * a.Di * a.Hi * a.Wi * cs;
            const int gz0 = bz * G::TZ - 2, gy = by * X3_TY - 2 + iy;
            const bool yok = (unsigned)gy < (unsigned)a.Hi;
            const int rs = a.Wi * cs, off0 = (gy * a.Wi + (col >> 2)) * cs;
#pragma unroll
            for (int k = 0; k < G::XPER; ++k) {
                const int gz = gz0 + k;
                const bool ok = yok && (unsigned)gz < (unsigned)a.Di;
                const float4 t = *reinterpret_cast<const float4*>(src + (ok ? off0 + gz * a.Hi * rs : 0));
                px[k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            XT::template issue_part<0, XT::PER>(px, a.x0, a.x1, a.C0, a.C1, chunk, b, bz * X3_TZ - 2, by * X3_TY - 2, bx * X3_TX - 2, a.Di, a.Hi, a.Wi, tid);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + k * NT;
            const int v = q >> 2, cq = q & 3;
            const int vx = v & 15, vy = (v >> 4) & 7, vz = v >> 7;
            const int oz = W8 ? bz * G::TZ + vz + 2 * (vx >> 3) : bz * X3_TZ + vz, oy = by * X3_TY + vy, ox = W8 ? (vx & 7) : bx * X3_TX + vx;
            const int c = co0 + cq * 4;
            const bool ok = oz < a.Do && oy < a.Ho && ox < a.Wo && c < a.Cout;
            const size_t ov = ok ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
            const float4 t = *reinterpret_cast<const float4*>(a.dy + ov * a.Cout + (ok ? c : 0));
            pd[k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](const bool neg) {
        if constexpr (W8) {
            const int stid = tid < 384 ? tid : tid - 128;
            unsigned char* base = xt + (stid >> 5) * XW_XROW + (((stid & 31) >> 2) + 2) * 32 + (stid & 3) * 8;
#pragma unroll
            for (int k = 0; k < G::XPER; ++k) {
                u32x2 h, m, l;
                x3_split4(px[k], h, m, l);
                unsigned char* dst = base + k * XW_XPLANE;
                *reinterpret_cast<u32x2*>(dst) = h;
                *reinterpret_cast<u32x2*>(dst + XW_XPB) = m;
                *reinterpret_cast<u32x2*>(dst + 2 * XW_XPB) = l;
            }
        } else {
        const int r0 = tid / XT::COLS, col = tid - r0 * XT::COLS;
        if (r0 < XT::RPI) {
            unsigned char* base = xt + (col >> 2) * 32 + (col & 3) * 8;
#pragma unroll
            for (int k = 0; k < XT::PER; ++k) {
                const int row = r0 + k * XT::RPI;
                u32x2 h, m, l;
                x3_split4(px[k], h, m, l);
                unsigned char* dst = base + row * XW_XROW;
                *reinterpret_cast<u32x2*>(dst) = h;
                *reinterpret_cast<u32x2*>(dst + XW_XPB) = m;
                *reinterpret_cast<u32x2*>(dst + 2 * XW_XPB) = l;
            }
        }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = tid + k * NT;
            const int v = q >> 2, cq = q & 3;
            const int vx = v & 15, row = v >> 4;
            u32x2 h, m, l;
            // (odd bricks: -dy.  The split is symmetric -- RNE -- so the pieces of -v are the negated pieces of v, exactly.)
            const float4 dv = neg ? make_float4(-pd[k].x, -pd[k].y, -pd[k].z, -pd[k].w) : pd[k];
            x3_split4(dv, h, m, l);
            unsigned char* dst = dyt + (row * 16 + (vx ^ ((row & 1) << 2))) * 32 + cq * 8;
            *reinterpret_cast<u32x2*>(dst) = h;
            *reinterpret_cast<u32x2*>(dst + XW_DPB) = m;
            *reinterpret_cast<u32x2*>(dst + 2 * XW_DPB) = l;
        }
    };

    // SIGN ALTERNATION (round 6).  The accumulation of v_mfma_f32_16x16x32_bf16 is not sign-symmetric: over long chains its results
    // sit BELOW the exact sum by an amount that grows with the number of instructions -- measured on this kernel (N(0,1) operands,
    // 32 x 64 x 128 voxels, 16 -> 16): mean error -7.9e-7 of rms |dw|, the same for every tap, i.e. almost all of the kernel's 9.3e-7
    // rel-L2 against the fp32 MFMA's 5.7e-7, and 1.7e-6 against 4.9e-7 at 64 x 128 x 128 (profiles/r06_x3_wgrad_taps.txt,
    // r06_x3_adversarial_chain.txt; the single-instruction probe profiles/r06_mfma_round_probe.txt shows the in-lane 24-bit window,
    // which truncates toward zero -- the one-sided part is below what one instruction reveals).  Whatever its seat, it does not
    // follow the sign of the operands: every other brick therefore runs NEGATED -- -dy goes into LDS (exact: the split is symmetric)
    // and the accumulators change sign at the brick boundary (exact) -- so consecutive, statistically equal bricks carry the offset
    // with opposite signs.  After: mean +1.0e-7, rel-L2 4.9e-7 (0.86 x the fp32 MFMA's); results stay deterministic.
    bool neg = false;
    if constexpr (W8) {                                // the x halo columns (0, 1, 10, 11) of every tile row: zero, never staged
        for (int q = tid; q < 3 * G::IZ * X3_IY * 4 * 2; q += NT) {
            const int hf = q & 1, c = (q >> 1) & 3, row = (q >> 3) % (G::IZ * X3_IY), piece = q / (8 * G::IZ * X3_IY);
            *reinterpret_cast<u32x4*>(xt + piece * XW_XPB + row * XW_XROW + (c < 2 ? c : c + 8) * 32 + hf * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (split < a.nbrick) issue(split);
    for (int brick = split; brick < a.nbrick; brick += a.nsplit) {
        __syncthreads();                               // every wave is done reading the previous tiles
        commit(neg);
        __syncthreads();
        if (brick != split) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = -acc[t];
        }
        neg = !neg;
        auto plane = [&](const int zz) {
            // (priority alternation between the two waves of a SIMD, as in the convolution kernel: the matrix pipe goes to the older wave
            //  whenever both are ready, so without it waves 0-3 finish a brick early and waves 4-7 finish it alone at a single wave's rate)
            if ((wave >> 2) ^ zz) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            const int zx = zz * XW_XPLANE, zd = zz * (X3_TY * 16 * 32);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned char* p = pb[c] + zx;
                bf16x8 F[5][3], A[3], An[3];
#pragma unroll
                for (int r = 0; r < 5; ++r) xw_frag(F[r], p + r * XW_XROW, p + r * XW_XROW + 128, XW_XPB);
                xw_frag(An, pa_lo + zd, pa_hi + zd, XW_DPB);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) A[q] = An[q];
                    if (s + 1 < 4) xw_frag(An, pa_lo + zd + (s + 1) * 1024, pa_hi + zd + (s + 1) * 1024, XW_DPB);
                    __builtin_amdgcn_sched_barrier(0);
                    xw_mfma6(acc[c * 5 + 0], A, F[(2 * s + 0) % 5]);
                    xw_mfma6(acc[c * 5 + 1], A, F[(2 * s + 1) % 5]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < 4) {                   // the two window slots that just died take the row pairs 2 s + 5, 2 s + 6
                        xw_frag(F[(2 * s + 0) % 5], p + (2 * s + 5) * XW_XROW, p + (2 * s + 5) * XW_XROW + 128, XW_XPB);
                        xw_frag(F[(2 * s + 1) % 5], p + (2 * s + 6) * XW_XROW, p + (2 * s + 6) * XW_XROW + 128, XW_XPB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    xw_mfma6(acc[c * 5 + 2], A, F[(2 * s + 2) % 5]);
                    xw_mfma6(acc[c * 5 + 3], A, F[(2 * s + 3) % 5]);
                    xw_mfma6(acc[c * 5 + 4], A, F[(2 * s + 4) % 5]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (extra) {                               // (wave-uniform) tap (4, dy = wave, 4)
                const unsigned char* p = pe + zx;
                bf16x8 E[3], En[3], A[3], An[3];
                xw_frag(En, p, p + 128, XW_XPB);
                xw_frag(An, pa_lo + zd, pa_hi + zd, XW_DPB);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) { A[q] = An[q]; E[q] = En[q]; }
                    if (s + 1 < 4) {
                        xw_frag(An, pa_lo + zd + (s + 1) * 1024, pa_hi + zd + (s + 1) * 1024, XW_DPB);
                        xw_frag(En, p + (2 * s + 2) * XW_XROW, p + (2 * s + 2) * XW_XROW + 128, XW_XPB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    xw_mfma6(acc[15], A, E);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        plane(0);
        // The next brick's loads go out BETWEEN the two planes (round 6), not right behind the barrier: there both waves of a SIMD spend
        // their first ~1.2 K cycles on the 14 loads' address arithmetic with the matrix pipe idle; here the other wave of the SIMD is
        // multiplying.  Unconditional (past the end: this brick again): a branch around an issue merges two vmcnt states and hipcc
        // then waits for the younger one.
        __builtin_amdgcn_sched_barrier(0);
        issue(brick + a.nsplit < a.nbrick ? brick + a.nsplit : brick);
        __builtin_amdgcn_sched_barrier(0);
        plane(1);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!neg) {                                        // (`neg` was toggled after the last brick: false = that brick ran negated)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = -acc[t];
    }
    // lane holds dW[tap][ci = chunk*16 + i][co = co0 + 4*g + {0..3}]
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        int tap;
        if (t < 15) { const int q = 3 * wave + t / 5, dz = q / 5, dx = q - dz * 5; tap = (dz * 5 + t % 5) * 5 + dx; }
        else { if (!extra) continue; tap = (4 * 5 + wave) * 5 + 4; }
        float* dst = a.part + ((size_t)(split * 125 + tap) * a.CinP + chunk * 16 + i) * a.CoutP + co0 + g * 4;
        const f32x4 r = acc[t];
        *reinterpret_cast<float4*>(dst) = make_float4(r.x, r.y, r.z, r.w);
    }
}

// bricks of a volume: 2 x 8 x 16, or 4 x 8 x 8 where the volume is exactly 8 wide (X3G<true>)
inline int x3_conv_nbz(int D, int W) { return W == 8 ? ceil_div(D, X3G<true>::TZ) : ceil_div(D, X3_TZ); }
inline int x3_conv_bricks(int B, int D, int H, int W) { return B * x3_conv_nbz(D, W) * ceil_div(H, X3_TY) * ceil_div(W, X3_TX); }
struct X3WgradPlan { int nbz, nby, nbx, nbrick, nsplit, nblk; };
inline X3WgradPlan x3_plan_wgrad(int Cin, int Cout, int B, int D, int H, int W) {
    X3WgradPlan p{};
    p.nbz = x3_conv_nbz(D, W); p.nby = ceil_div(H, X3_TY); p.nbx = ceil_div(W, X3_TX);
    p.nbrick = B * p.nbz * p.nby * p.nbx;
    p.nblk = (Cin / 16) * (Cout / 16);
    p.nsplit = max(1, min(p.nbrick, ceil_div(256, p.nblk)));
    return p;
}
inline bool x3_wgrad_ok(int C0, int C1, int Cout, int B, int D, int H, int W) {
    if (C0 <= 0 || Cout <= 0 || (C0 & 15) || (C1 & 15) || (Cout & 15) || (W < 16 && W != 8)) return false;
    const X3WgradPlan p = x3_plan_wgrad(C0 + C1, Cout, B, D, H, W);
    // at least four bricks per workgroup: the first tile load is exposed (two of the narrow brick's, which are twice as deep)
    return p.nbrick >= (W == 8 ? 2 : 4) * p.nsplit;
}

// does the f32x3 kernel take this 5^3 stride-1 problem?  (whole 16-channel blocks on both sides, rows of >= 16 voxels, and enough
// (brick, cout block) items for one round of the chip -- if need be with the channel chunks split over several workgroups:
// partial slabs + splitk_reduce_kernel, the deep levels)
struct X3Plan { int ok, nks, cps; long items; };
inline X3Plan x3_plan_conv(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) {
    X3Plan p{0, 1, 0, 0};
    if (C0 <= 0 || Cy0 <= 0 || (C0 & 15) || (C1 & 15) || (Cy0 & 15) || (Cy1 & 15) || (W < 16 && W != 8)) return p;
    const int nch = (C0 + C1) / 16;
    p.items = (long)x3_conv_bricks(B, D, H, W) * ((Cy0 + Cy1) / 16);
    p.cps = nch;
    while (p.items * p.nks < 256 && p.cps % 2 == 0 && p.cps >= 4) { p.nks *= 2; p.cps /= 2; }      // at least two chunks per item
    p.ok = p.items * p.nks >= 192;
    return p;
}
inline bool x3_conv_ok(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) { return x3_plan_conv(C0, C1, Cy0, Cy1, B, D, H, W).ok != 0; }

}  // namespace

// conv_deep.h -- wide-channel implicit-GEMM kernel for the DEEP levels of the bf16-storage 5^3 convolution (round 4).
//
// Replaces, for 32^3 64->64 / 16^3 128->128 / 8^3 256->256 and their two-source / backward-data relatives, the generic
// conv5_bf16_kernel of conv_kernels.h (reference: layers2.py:59-63 called from networks.py:280-282,307-322 at levels 2..4).
// Those launches are skinny GEMMs -- M = 512 .. 32 768 voxels against K = 8 000 .. 32 000 -- and the generic kernel ran them
// at 0.12-0.33 of the bf16 MFMA peak: one four-wave workgroup per CU, the filter staged plane by plane through LDS (global ->
// registers -> ds_write, two barriers per plane), 1.5 KB of LDS reads per MFMA; with the MFMAs removed it still took 31 of
// its 42 us.  This kernel removes that pipeline instead of tuning it:
//
//   * a workgroup (8 waves) owns one 4x8x8 brick x one 32-cout block x a range of 16-cin chunks; EVERY wave holds the whole
//     brick -- 8 accumulator tiles D[32 cout][32 voxels] of v_mfma_f32_32x32x16_bf16 = 128 registers -- and the waves split K:
//     a work unit is (chunk, dy, dx) with all five dz;
//   * the FILTER never touches LDS: a wave's A fragments are private to it (nobody else multiplies that (chunk, tap)), so they
//     stream global -> VGPR in the packed fragment order (one coalesced 1 KB load per tap), a whole unit ahead of their use.
//     No filter planes, no per-plane barriers, no ds_write of weights;
//   * z-sliding B reuse: tile plane p of the brick + halo feeds output plane z = p - dz for every dz, so a unit reads
//     8 planes x 2 y-halves = 16 B fragments for 40 MFMAs: 0.4 KB of LDS per MFMA (generic: 1.5, row-pair kernel: 0.8);
//   * the tile (brick + halo of one chunk: 36 KB) is staged through registers into two buffers; per chunk every wave does
//     three units (dy, dx) = wave, 8 + wave, 16 + wave and one fifth (one dz) of the 25th, so the single barrier per chunk
//     finds the waves together; straight-line code per chunk, every prefetch issued unconditionally (see the kernel);
//   * at the end the eight partial bricks meet in LDS (two rounds of 4 tiles x 8 waves x 4 KB = 128 KB), every wave sums half
//     a tile per round in a fixed order (deterministic) and runs the ordinary epilogue on it: bias, accumulate, one RNE rounding, statistics,
//     or the fp32 split-K slab when the chunk range is split over workgroups.
#pragma once
#include "conv_kernels.h"
#include <type_traits>

namespace {

// Experiment build -DDEEP_STAMPS (profiles/probes/deep_probe.hip): s_memtime stamps of the eight waves of one workgroup at the
// phase boundaries of the kernel; -DDEEP_NO_A / -DDEEP_NO_B / -DDEEP_NO_MFMA: timing-only ablations (results are wrong).
#ifdef DEEP_STAMPS
#ifndef DEEP_STAMP_BLOCK
#define DEEP_STAMP_BLOCK 1
#endif
#ifndef DEEP_STAMP_CHUNK
#define DEEP_STAMP_CHUNK 0
#endif
static __device__ long long* g_deep_stamps = nullptr;
#define DEEP_STAMP(k) do { if (blockIdx.x == DEEP_STAMP_BLOCK && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && g_deep_stamps) \
        g_deep_stamps[wave * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DEEP_STAMP(k) do {} while (0)
#define DEEP_STAMP_CHUNK 0
#endif

struct DeepPlan { int use, nbz, nby, nbx, ncob, nsplit, cps; };

// when the deep kernel takes a bf16-storage 5^3 convolution, and how it is cut: bricks of 4x8x8, 32-cout blocks, K split over
// workgroups until about one workgroup per CU exists
inline DeepPlan plan_conv_deep(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W, bool ignore_env = false) {
    DeepPlan p{};
    const int Cin = C0 + C1, Cout = Cy0 + Cy1;
    const char* env = getenv("VNET_BF16_DEEP");          // 0: off (the generic kernels; read per call: tests and A/B runs flip it)
    if (!ignore_env && env && atoi(env) == 0) return p;
    if ((Cout & 31) || (C0 & 15) || (C1 & 15) || (Cy0 & 3) || (Cy1 & 3) || Cin < 16) return p;
    if (conv_bf16_use_c16(Cin, Cout, C0, C1, Cy0, Cy1, B, D, H, W)) return p;
    const int nchunks = Cin / 16;
    p.nbz = ceil_div(D, 4); p.nby = ceil_div(H, 8); p.nbx = ceil_div(W, 8);
    p.ncob = Cout / 32;
    const long nwg0 = (long)B * p.nbz * p.nby * p.nbx * p.ncob;
    if (nwg0 > 512) return p;                        // enough bricks for the persistent row-pair / generic kernels
    if (conv_bf16_use_r32(Cout, Cy0, Cy1, B, D, H, W) && nwg0 >= 256) {
        Bf16Plan g = plan_conv_bf16(Cin, Cout, B, D, H, W);
        if (g.nsplit * g.nz == 1) return p;          // the row-pair kernel takes it
    }
    const char* tenv = ignore_env ? nullptr : getenv("VNET_BF16_DEEP_TARGET");      // workgroups the K split aims for (tests: 1 = no split)
    const int tgt = tenv ? atoi(tenv) : 256;
    int ns = (int)max(1l, min((long)nchunks, (tgt + nwg0 - 1) / nwg0));
    p.cps = ceil_div(nchunks, ns);
    p.nsplit = ceil_div(nchunks, p.cps);
    p.use = 1;
    return p;
}

template <bool STATS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_bf16_deep_kernel(ConvArgs a) {
    using G = Bf16Geom<4, 8, 8>;
    constexpr int NT = 512, NW = 8, NBUF = 2;
    constexpr int SCRATCH = 4 * NW * 4096;                              // one reduction round: 4 tiles x 8 waves x 4 KB
    constexpr int MAIN = (NBUF * G::TILE_BYTES > SCRATCH) ? NBUF * G::TILE_BYTES : SCRATCH;
    constexpr int PSTRIDE = G::IY * G::IX * 16;                         // bytes between two z planes of a cin-half plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using XH = XTileH<G::IZ, G::IY, G::IX, NT>;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DEEP_STAMP(0);
    const int p32 = lane & 31, half = lane >> 5;
    // voxel of this lane inside a 32-voxel subtile (4 rows of 8 along x; the bank-conflict-free lane order of conv5_bf16_kernel)
    const bool ga = p32 < 4 || (p32 >= 12 && p32 < 16) || (p32 >= 20 && p32 < 28);
    const int jq = ga ? (p32 < 4 ? p32 : p32 < 16 ? p32 - 8 : p32 - 12) : (p32 < 12 ? p32 - 4 : p32 < 20 ? p32 - 8 : p32 - 16);
    const int q32 = ((jq >> 3) * 2 + (ga ? 0 : 1)) * 8 + (jq & 7);

    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    int brick = xcd_remap(blockIdx.x, nbrick);
    const int brick_id = brick;
    const int bx = brick % a.nbx; brick /= a.nbx;
    const int by = brick % a.nby; brick /= a.nby;
    const int bz = brick % a.nbz; const int b = brick / a.nbz;
    const int ncob = a.CoutP / 32;
    const int cob = blockIdx.y, co0 = cob * 32;
    const int c_begin = blockIdx.z * a.cps;
    const int ncl = min(a.nchunks, c_begin + a.cps) - c_begin;          // chunks of this workgroup

    const int gz0 = bz * 4 - 2, gy0 = by * 8 - 2, gx0 = bx * 8 - 2;
    const unsigned short* x0h = reinterpret_cast<const unsigned short*>(a.x0);
    const unsigned short* x1h = reinterpret_cast<const unsigned short*>(a.x1);
    unsigned char* dump = smem + MAIN + lane * 16;

    // B fragment base of (y half 0, plane 0, tap (0,0)): lane -> (cin half, row, x)
    const int boff0 = half * G::PLANE + ((q32 >> 3) * G::IX + (q32 & 7)) * 16;
    constexpr int YH = 4 * G::IX * 16;                                  // second y half: four tile rows further
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (size_t)cob * 64 + lane;
    const size_t wtap = (size_t)ncob * 64;                              // u32x4 units between two taps

    f32x16 acc[8];
    u32x4 tv[XH::PER];                                                  // tile prefetch: the next chunk, global -> registers
    u32x4 afx[5], afy[5], am;                                           // filter fragments: this unit / the next one / the tail unit

    // Every load of the main loop is issued UNCONDITIONALLY (past the end: the last chunk's filter again, the zero line for
    // the tile): a branch around an issue makes hipcc merge two counter states at the join and wait for the YOUNGER one,
    // i.e. s_waitcnt vmcnt(4..0) instead of (9..5) in front of the MFMAs -- the prefetch of the next unit then has to land
    // inside the first planes of this one (round-4 stamps: 72 % of the MFMA rate; without the filter loads 100 %).
    auto a_issue5 = [&](u32x4 (&f)[5], int c, int r) {                  // r = dy * 5 + dx
        const u32x4* src = wg + ((size_t)(c_begin + c) * 125 + r) * wtap;
#ifdef DEEP_A1
        f[0] = *(const __attribute__((address_space(1))) u32x4*)(src);
#elif !defined(DEEP_NO_A)
#pragma unroll
        for (int dz = 0; dz < 5; ++dz) f[dz] = *(const __attribute__((address_space(1))) u32x4*)(src + (size_t)dz * 25 * wtap);
#endif
    };
    // Tile staging, one 16-byte piece k = 0..4 of this thread at a time (thread = (tile row r0 of 21, x, cin half) column as in
    // XTileH; piece k = tile row r0 + 21 k).  Branch-free -- masked lanes load the zero line, masked stores go to a dump slot --
    // so that the pieces can sit between the MFMAs of a unit (round-4 stamps: the XTileH call + commit + the waits hipcc put
    // behind its interior / boundary branch cost 3.5 K cycles per chunk, a third of the main loop).
    const int t_r0 = tid / XH::COLS, t_col = tid - t_r0 * XH::COLS;
    const int t_ix = t_col >> 1, t_hf = t_col & 1;
    const int t_gx = gx0 + t_ix;
    const bool t_colok = t_r0 < XH::RPI && (unsigned)t_gx < (unsigned)a.Wi;
    const int t_lds0 = t_hf * G::PLANE + (t_r0 * G::IX + t_ix) * 16;     // + k * RPI * IX * 16
    auto t_issue_k = [&](int c, int k) {
        const int cg = c_begin + c;
        const bool live = c < ncl;                                      // (uniform) past the last chunk: the zero line
        const bool s0 = cg * 16 < a.C0;                                 // (uniform) which source the chunk lies in
        const unsigned short* src = s0 ? x0h + cg * 16 : x1h + (cg * 16 - a.C0);
        const int Cs = s0 ? a.C0 : a.C1;
        const int row = t_r0 + k * XH::RPI;
        const int iz = (row * 5462) >> 16, iy = row - iz * G::IY;      // row / 12 for row < 192
        const int gz = gz0 + iz, gy = gy0 + iy;
#ifdef DEEP_TILE_ZERO
        const bool ok = live && t_colok && row < XH::ROWS && (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && c < 1;
#else
        const bool ok = live && t_colok && row < XH::ROWS && (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi;
#endif
        const long long vox = ((long long)(b * a.Di + gz) * a.Hi + gy) * a.Wi + t_gx;
        tv[k] = load16_or_zero(src + (ok ? vox * Cs + t_hf * 8 : 0), ok);
    };
    auto t_commit_k = [&](int c, int k) {
        const int row = t_r0 + k * XH::RPI;
        const bool ok = t_r0 < XH::RPI && row < XH::ROWS;
        unsigned char* dst = ok ? smem + (c & 1) * G::TILE_BYTES + t_lds0 + k * (XH::RPI * G::IX * 16) : dump;
        *reinterpret_cast<u32x4*>(dst) = tv[k];
    };
    // one unit = one (dy, dx) with all five dz: 16 B fragments (8 planes x 2 y halves) feed 40 MFMAs.  Fragment ring two planes
    // deep (the outer planes carry only 2 MFMAs = 64 cycles, less than an LDS round trip), issue points pinned: hipcc
    // otherwise sinks every read next to its first use (read -> lgkmcnt(0) -> MFMA)
    auto unit = [&](const u32x4 (&f)[5], int c, int r) {
        const int dy = r / 5, dx = r - dy * 5;
        const unsigned char* tp = smem + (c & 1) * G::TILE_BYTES + boff0 + (dy * G::IX + dx) * 16;
        bf16x8 bb[3][2];
#ifdef DEEP_NO_B
#pragma unroll
        for (int p = 0; p < 3; ++p) { bb[p][0] = __builtin_bit_cast(bf16x8, f[p]); bb[p][1] = __builtin_bit_cast(bf16x8, f[p + 1]); }
        asm volatile("" :: "v"(tp));
#else
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bb[p][0] = *reinterpret_cast<const bf16x8*>(tp + p * PSTRIDE);
            bb[p][1] = *reinterpret_cast<const bf16x8*>(tp + p * PSTRIDE + YH);
        }
#endif
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#ifndef DEEP_NO_B
            if (p + 2 < 8) {
                bb[(p + 2) % 3][0] = *reinterpret_cast<const bf16x8*>(tp + (p + 2) * PSTRIDE);
                bb[(p + 2) % 3][1] = *reinterpret_cast<const bf16x8*>(tp + (p + 2) * PSTRIDE + YH);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifdef DEEP_NO_MFMA
            asm volatile("" :: "v"(bb[p % 3][0]), "v"(bb[p % 3][1]), "v"(f[p < 5 ? p : 4]));
#else
#pragma unroll
            for (int dz = 0; dz < 5; ++dz) {
                const int z = p - dz;
                if (z < 0 || z > 3) continue;
#ifdef DEEP_A1
                const bf16x8 av = __builtin_bit_cast(bf16x8, f[0]);
#else
                const bf16x8 av = __builtin_bit_cast(bf16x8, f[dz]);
#endif
                acc[2 * z] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[p % 3][0], acc[2 * z], 0, 0, 0);
                acc[2 * z + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[p % 3][1], acc[2 * z + 1], 0, 0, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the 25th (dy, dx) = (4, 4) of a chunk, split by dz over waves 0..4: 8 B fragments, 8 MFMAs each (25 units do not divide
    // by 8 waves: with it every wave does 3 units + this per chunk, and the per-chunk barrier finds them together)
    const int dzm = min(wave, 4);
    auto tail_issue = [&](int c) {
#ifndef DEEP_NO_A
        am = *(const __attribute__((address_space(1))) u32x4*)(wg + ((size_t)(c_begin + c) * 125 + dzm * 25 + 24) * wtap);
#endif
    };
    auto tail = [&](int c) {
        if (wave < 5) {
            const unsigned char* tp = smem + (c & 1) * G::TILE_BYTES + boff0 + (4 * G::IX + 4) * 16 + wave * PSTRIDE;
            bf16x8 bt[4][2];
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                bt[z][0] = *reinterpret_cast<const bf16x8*>(tp + z * PSTRIDE);
                bt[z][1] = *reinterpret_cast<const bf16x8*>(tp + z * PSTRIDE + YH);
            }
            const bf16x8 av = __builtin_bit_cast(bf16x8, am);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                acc[2 * z] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bt[z][0], acc[2 * z], 0, 0, 0);
                acc[2 * z + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bt[z][1], acc[2 * z + 1], 0, 0, 0);
            }
        }
    };
    // chunk c: its tile is visible in buffer c & 1, its first unit's fragments are in flight in fx; leaves the next chunk's first
    // unit in fy (the buffers swap roles every chunk: three units each)
    // Staging of the tiles under the chunk: pieces {0,1}, {2,3}, {4} at the head of the three units, each right BEHIND that unit's
    // filter prefetch.  s_waitcnt vmcnt counts in order, so a tile load in flight holds up every wait for a filter fragment
    // issued after it; issued behind the prefetch of unit u + 1 the first wait that covers it is the one for unit u + 2, two
    // units (5-6 K cycles) later -- a tile piece takes 3-4 K cycles when every workgroup of the launch asks for its own at the
    // same moment (round-4 stamps), and spreading the pieces over the units thins that burst.  A piece is committed one chunk
    // after its issue (its buffer is free since the barrier that opened this chunk), then its registers take the next piece.
    auto stage = [&](int c, int k0, int k1) {
        if (c + 1 < ncl) {                                              // (uniform branch around stores: no load inside, the counters stay exact)
            for (int k = k0; k < k1; ++k) t_commit_k(c + 1, k);
        }
        for (int k = k0; k < k1; ++k) t_issue_k(c + 2, k);
    };
    auto chunk = [&](int c, u32x4 (&fx)[5], u32x4 (&fy)[5]) {
        const int cn = min(c + 1, ncl - 1);
        a_issue5(fy, c, 8 + wave);
        stage(c, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        unit(fx, c, wave);
        if (c == DEEP_STAMP_CHUNK) DEEP_STAMP(8);
        a_issue5(fx, c, 16 + wave);
        stage(c, 2, 4);
        __builtin_amdgcn_sched_barrier(0);
        unit(fy, c, 8 + wave);
        if (c == DEEP_STAMP_CHUNK) DEEP_STAMP(9);
        tail_issue(c);
        a_issue5(fy, cn, wave);
        stage(c, 4, 5);
        __builtin_amdgcn_sched_barrier(0);
        unit(fx, c, 16 + wave);
        if (c == DEEP_STAMP_CHUNK) DEEP_STAMP(10);
        tail(c);
        if (c == DEEP_STAMP_CHUNK) DEEP_STAMP(11);
        __syncthreads();                                                // chunk c + 1's tile (committed during this chunk) is visible
        if (c == DEEP_STAMP_CHUNK) DEEP_STAMP(13);
    };

    // prologue: chunk 0 into buffer 0 (its loads go out before anything else), chunk 1 on its way in registers, the first unit's
    // filter fragments in flight.  The loads that stay in flight into the loop are issued in the order the loop itself leaves
    // them at a chunk boundary -- tile pieces 0..3, filter fragments of the first unit, tile piece 4 -- because hipcc merges the
    // counter state of the loop entry with that of the back edge and waits for the younger of the two: with the pieces issued
    // last, every commit of the loop waited vmcnt(9), i.e. for loads issued one unit earlier.
    __builtin_amdgcn_sched_barrier(0);
    DEEP_STAMP(1);
    u32x4 tv0[XH::PER];
#pragma unroll
    for (int k = 0; k < XH::PER; ++k) { t_issue_k(0, k); tv0[k] = tv[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) t_issue_k(1, k);
    a_issue5(afx, 0, wave);
    t_issue_k(1, 4);
    __builtin_amdgcn_sched_barrier(0);
    DEEP_STAMP(2);
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    {
        u32x4 keep[XH::PER];
#pragma unroll
        for (int k = 0; k < XH::PER; ++k) { keep[k] = tv[k]; tv[k] = tv0[k]; t_commit_k(0, k); tv[k] = keep[k]; }
    }
    __syncthreads();
    DEEP_STAMP(3);
    {
        int c = 0;
        for (; c + 1 < ncl; c += 2) {
            chunk(c, afx, afy);
            chunk(c + 1, afy, afx);
        }
        if (c < ncl) chunk(c, afx, afy);
    }

    // ---- the eight partial bricks meet in LDS.  Two rounds of four tiles (128 KB); in round rd wave w sums HALF a tile: registers
    // 8h .. 8h+7 (h = w & 1: cout groups 2h, 2h+1) of tile 4 rd + (w >> 1), in a fixed order (deterministic) ----
    DEEP_STAMP(4);
    const int hsel = wave & 1, tsel = wave >> 1;
    float fin[2][8];
    float* sc = reinterpret_cast<float*>(smem);
    // (bias of this wave's channels: in flight under the reduction)
    float4 bv[2];
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
        const int co = co0 + (2 * hsel + gg) * 8 + half * 4;
        bv[gg] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && !a.part) bv[gg] = *reinterpret_cast<const float4*>(a.bias + co);
    }
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        if (rd == 1) __syncthreads();                                   // (the main loop ended with a barrier) round 0's reads are done
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (t == tsel && (g >> 1) == hsel) continue;            // the part this wave sums itself
                const f32x4 v = {acc[rd * 4 + t][4 * g], acc[rd * 4 + t][4 * g + 1], acc[rd * 4 + t][4 * g + 2], acc[rd * 4 + t][4 * g + 3]};
                *reinterpret_cast<f32x4*>(sc + (((t * NW + wave) * 4 + g) * 64 + lane) * 4) = v;
            }
        }
        __syncthreads();
        if (rd == 0) DEEP_STAMP(5);
#pragma unroll
        for (int k = 0; k < 8; ++k) fin[rd][k] = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (w == wave) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        if (tt == tsel && hh == hsel) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) fin[rd][k] += acc[rd * 4 + tt][8 * hh + k];
                        }
            } else {
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sc + (((tsel * NW + w) * 4 + 2 * hsel + gg) * 64 + lane) * 4);
                    fin[rd][4 * gg] += v[0]; fin[rd][4 * gg + 1] += v[1]; fin[rd][4 * gg + 2] += v[2]; fin[rd][4 * gg + 3] += v[3];
                }
            }
        }
    }
    DEEP_STAMP(6);

    // ---- epilogue: this wave holds, for rd = 0, 1: voxel (tile 4 rd + tsel, q32), channels co0 + (2 hsel + gg) * 8 + 4 half + k ----
    size_t ovs[4]; int cos[4]; bool oks[4]; float e[4][4];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int tile = rd * 4 + tsel;
        const int oz = bz * 4 + (tile >> 1), oy = by * 8 + (tile & 1) * 4 + (q32 >> 3), ox = bx * 8 + (q32 & 7);
        const bool vok = oz < a.Do && oy < a.Ho && ox < a.Wo;
        const size_t ov = vok ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int u = rd * 2 + gg;
            oks[u] = vok; ovs[u] = ov; cos[u] = co0 + (2 * hsel + gg) * 8 + half * 4;
            e[u][0] = fin[rd][4 * gg] + bv[gg].x; e[u][1] = fin[rd][4 * gg + 1] + bv[gg].y;
            e[u][2] = fin[rd][4 * gg + 2] + bv[gg].z; e[u][3] = fin[rd][4 * gg + 3] + bv[gg].w;
        }
    }
    if (a.part) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (oks[u]) *reinterpret_cast<float4*>(a.part + blockIdx.z * a.part_stride + ovs[u] * a.CoutP + cos[u]) = make_float4(e[u][0], e[u][1], e[u][2], e[u][3]);
        DEEP_STAMP(7);
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (!oks[u]) cos[u] = 0;
    epilogue_b16_batch<STATS, 4>(a, ovs, cos, oks, e);
    DEEP_STAMP(7);
    if constexpr (STATS) {
        // e now holds what the batch-norm behind sees (rounded value + residual): per channel over this wave's 2 x 32 voxels, then
        // over the waves; a wave owns 16 of the block's 32 channels and writes zeros for the others
        __syncthreads();                                                // the reduction scratch is free again
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int gg = 0; gg < 2; ++gg)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v1 = (oks[gg] ? e[gg][k] : 0.f) + (oks[2 + gg] ? e[2 + gg][k] : 0.f);
                float v2 = (oks[gg] ? e[gg][k] * e[gg][k] : 0.f) + (oks[2 + gg] ? e[2 + gg][k] * e[2 + gg][k] : 0.f);
                v1 = half32_sum(v1); v2 = half32_sum(v2);
                if (p32 == 0) {
                    const int ch = (2 * hsel + gg) * 8 + half * 4 + k, other = ch ^ 16;
                    red[wave * 64 + ch] = v1; red[wave * 64 + 32 + ch] = v2;
                    red[wave * 64 + other] = 0.f; red[wave * 64 + 32 + other] = 0.f;
                }
            }
        __syncthreads();
        stats_row_write<NW, 32>(red, a.stats, (size_t)brick_id, co0, a.Cout, tid);
    }
}

int launch_conv_deep(const ConvArgs& a, const DeepPlan& p, hipStream_t st) {
    using G = Bf16Geom<4, 8, 8>;
    constexpr size_t main_bytes = (2 * G::TILE_BYTES > 4 * 8 * 4096) ? 2 * G::TILE_BYTES : 4 * 8 * 4096;
    const size_t lds = main_bytes + 64 * 16;
    dim3 grid(a.B * p.nbz * p.nby * p.nbx, p.ncob, p.nsplit);
    if (a.stats && p.nsplit == 1) {
        auto k = conv5_bf16_deep_kernel<true>;
        static unsigned long long attr_done = 0;
        if (int ae = ensure_lds(k, lds, attr_done)) return ae;
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    } else {
        auto k = conv5_bf16_deep_kernel<false>;
        static unsigned long long attr_done = 0;
        if (int ae = ensure_lds(k, lds, attr_done)) return ae;
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    }
    return (int)hipGetLastError();
}

}  // namespace

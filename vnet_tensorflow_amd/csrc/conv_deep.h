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
//   * the tiles (brick + halo, 36 KB per chunk) of up to four chunks = 64 input channels = one 128-byte line per voxel are
//     resident in LDS for a whole phase and loaded in whole lines; inside a phase there is no barrier and no staging at
//     all: per chunk every wave does three units (dy, dx) = wave, 8 + wave, 16 + wave and one dz of the 25th (which five
//     waves, rotates), straight-line code, filter prefetch two units ahead, every issue unconditional (see the kernel);
//   * at the end the eight partial bricks meet in LDS (two rounds of 4 tiles x 8 waves x 4 KB = 128 KB), every wave sums half
//     a tile per round in a fixed order (deterministic) and runs the ordinary epilogue on it (four rounds of 64 KB through two
//     alternating buffers were measured and are SLOWER, 8.8 K against 7.2 K cycles: the LDS serves one access at a time and
//     every extra barrier costs more than the overlap gains): bias, accumulate, one RNE rounding, statistics,
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
    if (!ignore_env && tuning().bf16_deep == 0) return p;          // option BF16_DEEP = 0: the generic kernels (tests and A/B runs flip it)
    if ((Cout & 31) || (C0 & 15) || (C1 & 15) || (Cy0 & 7) || (Cy1 & 7) || Cin < 16) return p;     // (16-byte epilogue accesses)
    if (conv_bf16_use_c16(Cin, Cout, C0, C1, Cy0, Cy1, B, D, H, W)) return p;
    const int nchunks = Cin / 16;
    p.nbz = ceil_div(D, 4); p.nby = ceil_div(H, 8); p.nbx = ceil_div(W, 8);
    p.ncob = Cout / 32;
    const long nwg0 = (long)B * p.nbz * p.nby * p.nbx * p.ncob;
    if (nwg0 > 512) return p;                        // enough bricks for the persistent row-pair / generic kernels
    if ((long long)B * D * H * W * max(C0, C1) >= (1ll << 31)) return p;       // (32-bit element offsets in its tile loads)
    if (conv_bf16_use_r32(Cout, Cy0, Cy1, B, D, H, W) && nwg0 >= 256) {
        Bf16Plan g = plan_conv_bf16(Cin, Cout, B, D, H, W);
        if (g.nsplit * g.nz == 1) return p;          // the row-pair kernel takes it
    }
    const int tgt = ignore_env ? 256 : max(1, tuning().bf16_deep_target);           // workgroups the K split aims for (tests: 1 = no split)
    int ns = (int)max(1l, min((long)nchunks, (tgt + nwg0 - 1) / nwg0));
    p.cps = ceil_div(nchunks, ns);
    p.nsplit = ceil_div(nchunks, p.cps);
    p.use = 1;
    return p;
}

template <bool STATS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_bf16_deep_kernel(ConvArgs a) {
    using G = Bf16Geom<4, 8, 8>;
    constexpr int NW = 8, NSLOT = 4;
    // a (chunk, cin half) plane is padded by one 16-byte unit: the eight planes of a phase then start 16 bytes apart modulo the
    // 256-byte bank row, so the eight lanes of a ds_write_b128 cycle -- eight pieces of ONE voxel, see load_pass -- hit
    // eight different bank quads (unpadded: the same one, 8-way conflict)
    constexpr int PLANE = G::PLANE + 16, TILE = 2 * PLANE;
    constexpr int SCRATCH = 4 * NW * 4096;                              // one reduction round: 4 tiles x 8 waves x 4 KB
    constexpr int MAIN = (NSLOT * TILE > SCRATCH) ? NSLOT * TILE : SCRATCH;
    constexpr int PSTRIDE = G::IY * G::IX * 16;                         // bytes between two z planes of a cin-half plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DEEP_STAMP(0);
    const int p32 = lane & 31, half = lane >> 5;
    // voxel of this lane inside a 32-voxel subtile (4 rows of 8 along x; the bank-conflict-free lane order of conv5_bf16_kernel)
    const bool ga = p32 < 4 || (p32 >= 12 && p32 < 16) || (p32 >= 20 && p32 < 28);
    const int jq = ga ? (p32 < 4 ? p32 : p32 < 16 ? p32 - 8 : p32 - 12) : (p32 < 12 ? p32 - 4 : p32 < 20 ? p32 - 8 : p32 - 16);
    const int q32 = ((jq >> 3) * 2 + (ga ? 0 : 1)) * 8 + (jq & 7);

    // (grid: bricks x cout blocks x K splits.  Putting all workgroups of one filter slice on one XCD -- a 1-D grid decoded per XCD --
    // was measured: no gain in the step, +1 K cycles of index arithmetic in every prologue; dropped.)
    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    const int ncob = a.CoutP / 32;
    int brick = xcd_remap(blockIdx.x, nbrick);
    const int bsplit = blockIdx.z;
    const int brick_id = brick;
    const int bx = brick % a.nbx; brick /= a.nbx;
    const int by = brick % a.nby; brick /= a.nby;
    const int bz = brick % a.nbz; const int b = brick / a.nbz;
    const int cob = blockIdx.y, co0 = cob * 32;
    const int c_begin = bsplit * a.cps;
    const int ncl = min(a.nchunks, c_begin + a.cps) - c_begin;          // chunks of this workgroup

    const int gz0 = bz * 4 - 2, gy0 = by * 8 - 2, gx0 = bx * 8 - 2;
    const unsigned short* x0h = reinterpret_cast<const unsigned short*>(a.x0);
    const unsigned short* x1h = reinterpret_cast<const unsigned short*>(a.x1);
    unsigned char* dump = smem + MAIN + lane * 16;

    // B fragment base of (y half 0, plane 0, tap (0,0)): lane -> (cin half, row, x)
    const int boff0 = half * PLANE + ((q32 >> 3) * G::IX + (q32 & 7)) * 16;
    constexpr int YH = 4 * G::IX * 16;                                  // second y half: four tile rows further
    // filter fragments: a UNIFORM base (this cout block) + a uniform 32-bit unit offset per (chunk, tap) + the lane -- the
    // saddr form of global_load; with the lane folded into a 64-bit vector pointer every fragment address was a 64-bit vector
    // multiply-add (1.5 K cycles for the ten loads of the prologue: round-4 stamps)
    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wp) + (size_t)cob * 64;
    const unsigned wtap = (unsigned)ncob * 64u;                         // u32x4 units between two taps (whole filter < 2^32 units)

    f32x16 acc[8];
    u32x4 afx[5], afy[5], afz[5], am;                                   // filter fragments: a ring of three units + the tail unit's

    // Every load of the main loop is issued UNCONDITIONALLY (past the end: the last chunk's filter again): a branch around an
    // issue makes hipcc merge two counter states at the join and wait for the YOUNGER one, i.e. s_waitcnt vmcnt(4..0) instead
    // of the exact count in front of the MFMAs -- the prefetch then has to land inside the first planes of the unit that
    // issued it (round-4 stamps: 72 % of the MFMA rate; without the filter loads 100 %).
    auto a_issue5 = [&](u32x4 (&f)[5], int c, int r) {                  // c: chunk of this workgroup, r = dy * 5 + dx
        const unsigned u0 = ((unsigned)(c_begin + min(c, ncl - 1)) * 125u + (unsigned)r) * wtap;
#ifdef DEEP_A1
        f[0] = *(const __attribute__((address_space(1))) u32x4*)(wbase + u0 + lane);
#elif !defined(DEEP_NO_A)
#pragma unroll
        for (int dz = 0; dz < 5; ++dz) f[dz] = *(const __attribute__((address_space(1))) u32x4*)(wbase + (u0 + (unsigned)dz * 25u * wtap) + lane);
#endif
    };

    // ---- tile staging.  A PHASE = up to four chunks (64 input channels = one 128-byte line per voxel), all resident in LDS for
    // the whole phase.  The generic kernels (and this one until its third version) staged one chunk at a time: 32 bytes of
    // every voxel's line per pass, i.e. every line of the brick + halo crossed the L2 -> L1 path once per chunk; those loads,
    // queued in order in front of the filter prefetch, held the main loop at 70 % of the MFMA rate (round-4 stamps: 47 K cycles
    // for 32.8 K of MFMAs at 32^3 64->64; 38.8 K now).  Here a pass moves 2^LG 16-byte pieces per voxel -- whole lines when a
    // phase has four chunks -- lanes in (voxel, piece) order: 16 lanes read two whole lines (with the lanes in (piece, voxel)
    // order the texture path saw 8 lines per 16 lanes and the pass took 11 K instead of 5 K cycles), and the piece goes to
    // its (chunk, cin half) plane.  Staged through two register batches of six; nothing else runs meanwhile.
    // FIRST (the first phase: the accumulators are not live yet): every load of the pass in flight at once; later phases: two
    // register batches of six.
    auto load_pass = [&](auto lgc, auto firstc, const unsigned short* src, int Cs, int slot0) {
        constexpr int LG = decltype(lgc)::value, NP = 1 << LG, VPI = 64 >> LG;      // pieces per voxel, voxels per instruction
        constexpr int NINST = (G::NV * NP + 63) / 64, NI = (NINST + NW - 1) / NW;   // wave instructions of the pass, per wave
        constexpr int NB = decltype(firstc)::value ? NI : 6;
        int vsub = lane >> LG;
        // (opaque per call: what follows depends only on the thread, and hipcc would otherwise hoist the address arithmetic of all
        // eighteen pieces out of the phase loop and keep it in registers for the whole kernel)
        asm volatile("" : "+v"(vsub));
        const int piece = lane & (NP - 1);
        const int ldsp = (slot0 + (piece >> 1)) * TILE + (piece & 1) * PLANE;
        // tile coordinates of this thread's first voxel (v0 < 64 * ... : one small division), then compile-time steps of VPI * NW
        // voxels with at most one carry per axis; 32-bit element offsets (the launcher vouches for < 2^31 elements per source):
        // ~20 VALU per load.  (The first version recomputed v / 144, v / 12 and a 64-bit offset per load, ~50 VALU: the
        // address arithmetic of the 18 loads, not the memory, was most of the 9.9 K cycles of the tile load.)
        const int v0 = wave * VPI + vsub;                               // < NW * VPI <= 256
        const int iz0 = v0 / (G::IY * G::IX), r0 = v0 - iz0 * (G::IY * G::IX), iy0 = (r0 * 171) >> 11, ix0 = r0 - iy0 * G::IX;   // r0 < 144: r0 / 12
        const unsigned sbase = (unsigned)b * (unsigned)(a.Di * a.Hi * a.Wi);
        auto issue = [&](u32x4 (&t)[NB], int k0) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k0 + k >= NI) break;
                const int dv = (k0 + k) * VPI * NW;                     // compile-time after unrolling
                const int dzk = dv / (G::IY * G::IX), drk = dv - dzk * (G::IY * G::IX), dyk = drk / G::IX, dxk = drk - dyk * G::IX;
                int ix = ix0 + dxk, iy = iy0 + dyk, iz = iz0 + dzk;
                const int cx = ix >= G::IX; ix -= cx * G::IX; iy += cx;
                const int cy = iy >= G::IY; iy -= cy * G::IY; iz += cy;
                const int gz = gz0 + iz, gy = gy0 + iy, gx = gx0 + ix;
                const bool ok = iz < G::IZ && (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi && (unsigned)gx < (unsigned)a.Wi;
                const unsigned vox = sbase + (unsigned)((gz * a.Hi + gy) * a.Wi + gx);
                t[k] = load16_or_zero(src + (ok ? vox * (unsigned)Cs + (unsigned)(piece * 8) : 0u), ok);
            }
        };
        auto commit = [&](const u32x4 (&t)[NB], int k0) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k0 + k >= NI) break;
                const int v = v0 + (k0 + k) * VPI * NW;
                *reinterpret_cast<u32x4*>(v < G::NV ? smem + ldsp + v * 16 : dump) = t[k];
            }
        };
        u32x4 ta[NB], tb[NB];
        issue(ta, 0);
#pragma unroll
        for (int k0 = 0; k0 < NI; k0 += 2 * NB) {
            if (k0 + NB < NI) issue(tb, k0 + NB);
            commit(ta, k0);
            if (k0 + 2 * NB < NI) issue(ta, k0 + 2 * NB);
            if (k0 + NB < NI) commit(tb, k0 + NB);
        }
    };
    // chunks [c0, c1) of this workgroup -> slots 0 .. c1 - c0 - 1, in groups of 4 / 2 / 1 chunks that stay inside one source tensor
    auto load_phase = [&](auto firstc, int c0, int c1) {
        int c = c0;
        while (c < c1) {
            const int cg = c_begin + c;
            const bool s0 = cg * 16 < a.C0;
            const unsigned short* src = s0 ? x0h + cg * 16 : x1h + (cg * 16 - a.C0);
            const int Cs = s0 ? a.C0 : a.C1;
            const int room = min(c1 - c, ((s0 ? a.C0 : a.C0 + a.C1) - cg * 16) / 16);     // chunks left in this phase and in this source
            if (room >= 4) { load_pass(std::integral_constant<int, 3>{}, firstc, src, Cs, c - c0); c += 4; }
            else if (room >= 2) { load_pass(std::integral_constant<int, 2>{}, firstc, src, Cs, c - c0); c += 2; }
            else { load_pass(std::integral_constant<int, 1>{}, firstc, src, Cs, c - c0); c += 1; }
        }
    };

    // one unit = one (dy, dx) with all five dz: 16 B fragments (8 planes x 2 y halves) feed 40 MFMAs.  The fragment ring is
    // two planes deep (the outer planes carry only 2 MFMAs = 64 cycles, less than an LDS round trip) and runs THROUGH the
    // units: planes 6 and 7 of a unit issue planes 0 and 1 of the next one (a unit that starts by reading its first fragments
    // waits an LDS round trip with the matrix pipe idle, twelve times per phase).  24 planes per chunk = 0 mod 3, so the ring
    // position of a unit's first plane is a compile-time constant (0, 2, 1).  Issue points pinned: hipcc otherwise sinks every
    // read next to its first use (read -> lgkmcnt(0) -> MFMA).
    bf16x8 bb[3][2];
    auto tile_ptr = [&](int slot, int r) {
        const int dy = r / 5, dx = r - dy * 5;
        return smem + slot * TILE + boff0 + (dy * G::IX + dx) * 16;
    };
    auto b_read = [&](bf16x8 (&dst)[2], const unsigned char* tp, int p) {
        dst[0] = *reinterpret_cast<const bf16x8*>(tp + p * PSTRIDE);
        dst[1] = *reinterpret_cast<const bf16x8*>(tp + p * PSTRIDE + YH);
    };
    auto unit = [&](auto basec, const u32x4 (&f)[5], const unsigned char* tp, const unsigned char* tp_next) {
        constexpr int BASE = decltype(basec)::value;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#ifndef DEEP_NO_B
            if (p + 2 < 8) b_read(bb[(BASE + p + 2) % 3], tp, p + 2);
            else b_read(bb[(BASE + p + 2) % 3], tp_next, p + 2 - 8);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifdef DEEP_NO_MFMA
            asm volatile("" :: "v"(bb[(BASE + p) % 3][0]), "v"(bb[(BASE + p) % 3][1]), "v"(f[p < 5 ? p : 4]));
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
                acc[2 * z] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[(BASE + p) % 3][0], acc[2 * z], 0, 0, 0);
                acc[2 * z + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[(BASE + p) % 3][1], acc[2 * z + 1], 0, 0, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the 25th (dy, dx) = (4, 4) of a chunk is split by dz over five waves (8 B fragments, 8 MFMAs each); which five rotates with
    // the chunk, and there is no barrier inside a phase, so the uneven share evens out over the chunks
    auto tail_dz = [&](int c) { return (wave - (c_begin + c)) & 7; };
    auto tail_issue = [&](int c) {
#ifndef DEEP_NO_A
        am = *(const __attribute__((address_space(1))) u32x4*)(wbase + ((unsigned)(c_begin + min(c, ncl - 1)) * 125u + (unsigned)(min(tail_dz(c), 4) * 25 + 24)) * wtap + lane);
#endif
    };
    auto tail = [&](int slot, int c) {
        const int dz = tail_dz(c);
        if (dz < 5) {
            const unsigned char* tp = smem + slot * TILE + boff0 + (4 * G::IX + 4) * 16 + dz * PSTRIDE;
            bf16x8 bt[2][2];
            bt[0][0] = *reinterpret_cast<const bf16x8*>(tp);
            bt[0][1] = *reinterpret_cast<const bf16x8*>(tp + YH);
            const bf16x8 av = __builtin_bit_cast(bf16x8, am);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                if (z + 1 < 4) {
                    bt[(z + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(tp + (z + 1) * PSTRIDE);
                    bt[(z + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(tp + (z + 1) * PSTRIDE + YH);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[2 * z] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bt[z & 1][0], acc[2 * z], 0, 0, 0);
                acc[2 * z + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bt[z & 1][1], acc[2 * z + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // chunk c (tile in `slot`): its units (dy, dx) = wave, 8 + wave, 16 + wave; fragments of the first two are in flight in
    // afx / afy on entry, the prefetch runs two units ahead (three buffers = the three units of a chunk: no role swapping)
    // (slot_next: where the next chunk's tile lies -- the last chunk of a phase points at its own slot: those reads are dropped)
    auto chunk = [&](int slot, int c, int slot_next) {
        const unsigned char* t0 = tile_ptr(slot, wave);
        const unsigned char* t1 = tile_ptr(slot, 8 + wave);
        const unsigned char* t2 = tile_ptr(slot, 16 + wave);
        a_issue5(afz, c, 16 + wave);
        __builtin_amdgcn_sched_barrier(0);
        unit(std::integral_constant<int, 0>{}, afx, t0, t1);
        a_issue5(afx, c + 1, wave);
        tail_issue(c);
        __builtin_amdgcn_sched_barrier(0);
        unit(std::integral_constant<int, 2>{}, afy, t1, t2);
        a_issue5(afy, c + 1, 8 + wave);
        __builtin_amdgcn_sched_barrier(0);
        unit(std::integral_constant<int, 1>{}, afz, t2, tile_ptr(slot_next, wave));
        tail(slot, c);
    };
    auto phase = [&](int c0, int c1) {
        b_read(bb[0], tile_ptr(0, wave), 0);
        b_read(bb[1], tile_ptr(0, wave), 1);
        for (int c = c0; c < c1; ++c) chunk(c - c0, c, c + 1 < c1 ? c - c0 + 1 : c - c0);
    };

    __builtin_amdgcn_sched_barrier(0);
    DEEP_STAMP(1);
    // first phase: the filter fragments of the first two units go out first, then the whole tile; the accumulators are cleared
    // while the loads fly
    a_issue5(afx, 0, wave);
    a_issue5(afy, 0, 8 + wave);
    DEEP_STAMP(2);
    load_phase(std::true_type{}, 0, min(ncl, NSLOT));
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    __syncthreads();
    DEEP_STAMP(3);
    phase(0, min(ncl, NSLOT));
    for (int c0 = NSLOT; c0 < ncl; c0 += NSLOT) {
        const int c1 = min(ncl, c0 + NSLOT);
        __syncthreads();                                                // every wave is done with the previous phase's tiles
        load_phase(std::false_type{}, c0, c1);
        a_issue5(afx, c0, wave);                                        // (behind the staging: its register batches need the room)
        a_issue5(afy, c0, 8 + wave);
        __syncthreads();
        phase(c0, c1);
    }
    __syncthreads();                                                    // the tiles are dead: the reduction scratch may overwrite them

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

    // ---- epilogue: this wave holds, for rd = 0, 1: voxel (tile 4 rd + tsel, q32), channels co0 + (2 hsel + gg) * 8 + 4 half + k.
    // Lanes L and L + 32 hold the same voxel: they exchange (v_permlane32_swap) so that lane L keeps cout group 2 hsel and
    // lane L + 32 group 2 hsel + 1 -- EIGHT consecutive channels each: one 16-byte load / store per lane and voxel instead of
    // two 8-byte ones (the store path is issue-bound: round-4 stamps had 2.5-3.9 K cycles here for 32 bytes per lane) ----
    const int cb = co0 + (2 * hsel + half) * 8;                         // first of this lane's eight channels
    float e8[2][8]; size_t ov2[2]; bool ok2[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int tile = rd * 4 + tsel;
        const int oz = bz * 4 + (tile >> 1), oy = by * 8 + (tile & 1) * 4 + (q32 >> 3), ox = bx * 8 + (q32 & 7);
        ok2[rd] = oz < a.Do && oy < a.Ho && ox < a.Wo;
        ov2[rd] = ok2[rd] ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float bk0 = k == 0 ? bv[0].x : k == 1 ? bv[0].y : k == 2 ? bv[0].z : bv[0].w;
            const float bk1 = k == 0 ? bv[1].x : k == 1 ? bv[1].y : k == 2 ? bv[1].z : bv[1].w;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fin[rd][k] + bk0), __float_as_uint(fin[rd][4 + k] + bk1), false, false);
            e8[rd][k] = __uint_as_float(sw[0]);                         // lower half: own group 2 hsel; upper half: partner's group 2 hsel + 1
            e8[rd][4 + k] = __uint_as_float(sw[1]);                     // lower half: partner's part of group 2 hsel (+4..7); upper half: own
        }
    }
    if (a.part) {
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
            if (ok2[rd]) {
                float* dst = a.part + bsplit * a.part_stride + ov2[rd] * a.CoutP + cb;
                *reinterpret_cast<float4*>(dst) = make_float4(e8[rd][0], e8[rd][1], e8[rd][2], e8[rd][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(e8[rd][4], e8[rd][5], e8[rd][6], e8[rd][7]);
            }
        DEEP_STAMP(7);
        return;
    }
    {
        const bool iny0 = cb < a.Cy0;                                   // (Cy0 % 8 == 0: eight channels never straddle y0 / y1)
        unsigned short* yb = iny0 ? reinterpret_cast<unsigned short*>(a.y0) + cb : reinterpret_cast<unsigned short*>(a.y1) + (cb - a.Cy0);
        const int ycs = iny0 ? a.Cy0 : a.Cy1;
        u32x4 old[2], rr[2];
        if (a.accum) {                                                  // every load of the epilogue in flight before the first use
            const unsigned short* sb = a.accsrc ? reinterpret_cast<const unsigned short*>(a.accsrc) + cb : yb;
            const int scs = a.accsrc ? a.Cy0 : ycs;
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) old[rd] = *reinterpret_cast<const u32x4*>(sb + ov2[rd] * scs);
        }
        if constexpr (STATS) {
            if (a.res) {
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) rr[rd] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.res) + ov2[rd] * a.Cout + cb);
            }
        }
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            if (a.accum) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { e8[rd][2 * k] += bf_lo(old[rd][k]); e8[rd][2 * k + 1] += bf_hi(old[rd][k]); }
            }
            u32x4 pk;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[k] = pk_bf16(e8[rd][2 * k], e8[rd][2 * k + 1]);
            if constexpr (STATS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { e8[rd][2 * k] = bf_lo(pk[k]); e8[rd][2 * k + 1] = bf_hi(pk[k]); }
                if (a.res) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { e8[rd][2 * k] += bf_lo(rr[rd][k]); e8[rd][2 * k + 1] += bf_hi(rr[rd][k]); }
                }
            }
            if (ok2[rd]) *reinterpret_cast<u32x4*>(yb + ov2[rd] * ycs) = pk;
        }
    }
    DEEP_STAMP(7);
    if constexpr (STATS) {
        // e8 now holds what the batch-norm behind sees (rounded value + residual): per channel over this wave's 2 x 32 voxels (the 32
        // lanes of a half hold the same eight channels), then over the waves; a wave owns 16 of the block's 32 channels and
        // writes zeros for the others
        __syncthreads();                                                // the reduction scratch is free again
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v1 = (ok2[0] ? e8[0][k] : 0.f) + (ok2[1] ? e8[1][k] : 0.f);
            float v2 = (ok2[0] ? e8[0][k] * e8[0][k] : 0.f) + (ok2[1] ? e8[1][k] * e8[1][k] : 0.f);
            v1 = half32_sum(v1); v2 = half32_sum(v2);
            if (p32 == 0) {
                const int ch = (2 * hsel + half) * 8 + k, other = ch ^ 16;
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
    constexpr size_t main_bytes = 4 * (G::TILE_BYTES + 32);          // four padded tiles (> the 128 KB of a reduction round)
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

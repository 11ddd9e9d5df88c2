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
//     a work unit is (chunk, dy, dx) with all five dz, unit u belongs to wave u mod 8;
//   * the FILTER never touches LDS: a wave's A fragments are private to it (nobody else multiplies that (chunk, tap)), so they
//     stream global -> VGPR in the packed fragment order (one coalesced 1 KB load per tap), a whole unit ahead of their use.
//     No filter planes, no per-plane barriers, no ds_write of weights;
//   * z-sliding B reuse: tile plane p of the brick + halo feeds output plane z = p - dz for every dz, so a unit reads
//     8 planes x 2 y-halves = 16 B fragments for 40 MFMAs: 0.4 KB of LDS per MFMA (generic: 1.5, row-pair kernel: 0.8);
//   * the tile (brick + halo of one chunk: 36 KB) is staged through registers into a ring of three buffers; a new chunk is
//     committed just before the first round that needs it, so there is ONE barrier per chunk and every wave does exactly one
//     unit between two barriers (25 units per chunk do not divide by 8: lock-step rounds instead of per-chunk loops keep the
//     waves balanced).  With three buffers the commit of chunk c cannot overtake a straggler still reading chunk c-3;
//   * at the end the eight partial bricks meet in LDS (two rounds of 4 tiles x 8 waves x 4 KB = 128 KB), wave w sums tile w
//     in a fixed order (deterministic) and runs the ordinary epilogue on it: bias, accumulate, one RNE rounding, statistics,
//     or the fp32 split-K slab when the chunk range is split over workgroups.
#pragma once
#include "conv_kernels.h"

namespace {

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
    constexpr int NT = 512, NW = 8, NBUF = 3;
    constexpr int SCRATCH = 4 * NW * 4096;                              // one reduction round: 4 tiles x 8 waves x 4 KB
    constexpr int MAIN = (NBUF * G::TILE_BYTES > SCRATCH) ? NBUF * G::TILE_BYTES : SCRATCH;
    constexpr int PSTRIDE = G::IY * G::IX * 16;                         // bytes between two z planes of a cin-half plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using XH = XTileH<G::IZ, G::IY, G::IX, NT>;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int nunits = ncl * 25, nrounds = (nunits + NW - 1) / NW;

    const int gz0 = bz * 4 - 2, gy0 = by * 8 - 2, gx0 = bx * 8 - 2;
    const unsigned short* x0h = reinterpret_cast<const unsigned short*>(a.x0);
    const unsigned short* x1h = reinterpret_cast<const unsigned short*>(a.x1);
    unsigned char* dump = smem + MAIN + lane * 16;

    // B fragment base of (y half yh, plane 0, tap (0,0)): lane -> (cin half, row, x)
    const int boff0 = half * G::PLANE + ((q32 >> 3) * G::IX + (q32 & 7)) * 16;
    constexpr int YH = 4 * G::IX * 16;                                  // second y half: four tile rows further
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (size_t)cob * 64 + lane;
    const size_t wtap = (size_t)ncob * 64;                              // u32x4 units between two taps

    f32x16 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    u32x4 tv[XH::PER];                                                  // tile prefetch: the next chunk, global -> registers
    u32x4 af[2][5];                                                     // filter fragments of the current / the next unit

    auto a_issue = [&](u32x4 (&f)[5], int u) {
        const int c = u / 25, r = u - c * 25;                           // r = dy * 5 + dx
        const u32x4* src = wg + ((size_t)(c_begin + c) * 125 + r) * wtap;
#pragma unroll
        for (int dz = 0; dz < 5; ++dz) f[dz] = *(const __attribute__((address_space(1))) u32x4*)(src + (size_t)dz * 25 * wtap);
    };
    auto t_issue = [&](int c) {
        XH::template issue_part<0, XH::PER>(tv, x0h, x1h, a.C0, a.C1, c_begin + c, b, gz0, gy0, gx0, a.Di, a.Hi, a.Wi, tid);
    };
    auto t_commit = [&](int c) {
        bf16_tile_commit_h<G, XH, 0, XH::PER>(smem + (c % NBUF) * G::TILE_BYTES, dump, tv, tid);
    };
    // one unit: 16 B fragments (8 planes x 2 y halves) x the five dz taps they serve = 40 MFMAs
    auto unit = [&](const u32x4 (&f)[5], int u) {
        const int c = u / 25, r = u - c * 25;
        const int dy = r / 5, dx = r - dy * 5;
        const unsigned char* tp = smem + (c % NBUF) * G::TILE_BYTES + boff0 + (dy * G::IX + dx) * 16;
        // fragment ring two planes deep (the outer planes carry only 2 MFMAs = 64 cycles, less than an LDS round trip),
        // issue points pinned: hipcc otherwise sinks every read next to its first use (read -> lgkmcnt(0) -> MFMA)
        bf16x8 bb[3][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bb[p][0] = *reinterpret_cast<const bf16x8*>(tp + p * PSTRIDE);
            bb[p][1] = *reinterpret_cast<const bf16x8*>(tp + p * PSTRIDE + YH);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p + 2 < 8) {
                bb[(p + 2) % 3][0] = *reinterpret_cast<const bf16x8*>(tp + (p + 2) * PSTRIDE);
                bb[(p + 2) % 3][1] = *reinterpret_cast<const bf16x8*>(tp + (p + 2) * PSTRIDE + YH);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dz = 0; dz < 5; ++dz) {
                const int z = p - dz;
                if (z < 0 || z > 3) continue;
                const bf16x8 av = __builtin_bit_cast(bf16x8, f[dz]);
                acc[2 * z] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[p % 3][0], acc[2 * z], 0, 0, 0);
                acc[2 * z + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bb[p % 3][1], acc[2 * z + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: chunk 0 into buffer 0, chunk 1 on its way in registers, the first unit's filter fragments in flight
    t_issue(0);
    if (wave < nunits) a_issue(af[0], wave);
    __builtin_amdgcn_sched_barrier(0);
    t_commit(0);
    if (ncl > 1) t_issue(1);
    __syncthreads();
    int committed = 1;

    auto round = [&](int j, const u32x4 (&fc)[5], u32x4 (&fn)[5]) {
        const int ul = min(j * NW + NW - 1, nunits - 1);
        if (ul / 25 >= committed) {                                     // (uniform) a new chunk enters with this round
            t_commit(committed);
            ++committed;
            __syncthreads();
            if (committed < ncl) t_issue(committed);
        }
        const int u = j * NW + wave;
        if (u < nunits) {
            if (u + NW < nunits) a_issue(fn, u + NW);
            __builtin_amdgcn_sched_barrier(0);
            unit(fc, u);
        }
    };
    for (int j = 0; j < nrounds; j += 2) {
        round(j, af[0], af[1]);
        if (j + 1 < nrounds) round(j + 1, af[1], af[0]);
    }

    // ---- the eight partial bricks meet in LDS: wave w ends up with tile w (z = w / 2, y half = w & 1) ----
    f32x16 fin;
    float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        __syncthreads();                                                // main loop / previous round done with the LDS
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (wave == rd * 4 + t) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[rd * 4 + t][4 * g], acc[rd * 4 + t][4 * g + 1], acc[rd * 4 + t][4 * g + 2], acc[rd * 4 + t][4 * g + 3]};
                *reinterpret_cast<f32x4*>(sc + (((t * NW + wave) * 4 + g) * 64 + lane) * 4) = v;
            }
        }
        __syncthreads();
        if ((wave >> 2) == rd) {
            const int t = wave & 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[r] = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (w == wave) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        if (tt == t) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) fin[r] += acc[rd * 4 + tt][r];
                        }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(sc + (((t * NW + w) * 4 + g) * 64 + lane) * 4);
                        fin[4 * g] += v[0]; fin[4 * g + 1] += v[1]; fin[4 * g + 2] += v[2]; fin[4 * g + 3] += v[3];
                    }
                }
            }
        }
    }

    // ---- epilogue of tile `wave`: register r of a lane = cout co0 + 8*(r/4) + 4*half + r%4 of voxel (wave, q32) ----
    const int vz = wave >> 1, vy = (wave & 1) * 4 + (q32 >> 3), vx = q32 & 7;
    const int oz = bz * 4 + vz, oy = by * 8 + vy, ox = bx * 8 + vx;
    const bool vok = oz < a.Do && oy < a.Ho && ox < a.Wo;
    const size_t ov = vok ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) s1[g][k] = s2[g][k] = 0.f;
    if (a.part) {
        if (vok) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(a.part + blockIdx.z * a.part_stride + ov * a.CoutP + co0 + g * 8 + half * 4) =
                    make_float4(fin[4 * g], fin[4 * g + 1], fin[4 * g + 2], fin[4 * g + 3]);
        }
        return;
    }
    {
        size_t ovs[4]; int cos[4]; bool oks[4]; float e[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = co0 + g * 8 + half * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bv = make_float4(a.bias[co], a.bias[co + 1], a.bias[co + 2], a.bias[co + 3]);
            oks[g] = vok; ovs[g] = ov; cos[g] = vok ? co : 0;
            e[g][0] = fin[4 * g] + bv.x; e[g][1] = fin[4 * g + 1] + bv.y; e[g][2] = fin[4 * g + 2] + bv.z; e[g][3] = fin[4 * g + 3] + bv.w;
        }
        epilogue_b16_batch<STATS, 4>(a, ovs, cos, oks, e);
        if constexpr (STATS) {
            if (vok) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s1[g][k] += e[g][k]; s2[g][k] += e[g][k] * e[g][k]; }
            }
        }
    }
    if constexpr (STATS) {
        __syncthreads();                                                // the reduction scratch is free again
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s1[g][k] = half32_sum(s1[g][k]);
                s2[g][k] = half32_sum(s2[g][k]);
                if (p32 == 0) {
                    red[wave * 64 + g * 8 + half * 4 + k] = s1[g][k];
                    red[wave * 64 + 32 + g * 8 + half * 4 + k] = s2[g][k];
                }
            }
        __syncthreads();
        stats_row_write<NW, 32>(red, a.stats, (size_t)brick_id, co0, a.Cout, tid);
    }
}

int launch_conv_deep(const ConvArgs& a, const DeepPlan& p, hipStream_t st) {
    using G = Bf16Geom<4, 8, 8>;
    constexpr size_t main_bytes = (3 * G::TILE_BYTES > 4 * 8 * 4096) ? 3 * G::TILE_BYTES : 4 * 8 * 4096;
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

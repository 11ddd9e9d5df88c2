// conv_kernels.h -- kernel templates, planners and launchers of the convolution family (gfx950), shared by the translation
// units that instantiate them: conv_mfma.hip (fp32 + bf16-operand entry points) and conv_b16.hip (bf16-storage entry points).
#pragma once
#include "common.h"
#include <mutex>
#include <unordered_map>
#include <type_traits>
#include <vector>

namespace {

struct ConvArgs {
    const float* x0; const float* x1; int C0, C1, Cin;
    const float4* wp; const float* bias;
    float* y0; float* y1; int Cy0, Cy1, Cout;
    int B, Di, Hi, Wi, Do, Ho, Wo;
    int CQ;            // padded Cin / 4
    int CoutP;         // padded Cout (multiple of 16)
    int nchunks, cps;  // 16-channel chunks, chunks per K-split
    int nbz, nby, nbx; // bricks per axis
    int pad, padx;     // low-side SAME padding (z,y) and along x
    int vec_in, vec_out;
    float* part; size_t part_stride;  // split-K partials [split][vox][CoutP]
    int upO;           // UP: real output channels O (N' = 8*O)
    int nz;            // tap (dz) splits per K-split: deep levels have too few bricks to fill 256 CUs
    int accum;         // 1: y += result (backward-data into a gradient another consumer of the same tensor already wrote)
    const float* accsrc;   // bf16 kernels, accum: read the other gradient from HERE (y0's layout) instead of y0 -- out of place
    // batch-norm statistics of the output in the epilogue (round 2): per-workgroup partial sums of v = y (+ res) and v^2 per
    // channel, row [2][Cout] per brick (or per reduce block for split-K launches); the batch-norm that consumes y then
    // only runs its finalize.  res: optional residual that is added in front of that batch-norm (networks.py:318).
    float* stats; const float* res;
    int in4;           // bf16 storage, 16-cout kernel: the single source is the network input, 4 real channels zero-padded to 8 (IN4)
};

// Experiment build -DVNET_STAMPS (profiles/build_stamps.sh): s_memtime stamps around the phases of four consecutive brick steps of
// one workgroup of the persistent bf16 kernels, written to the buffer set by vnet_debug_set_stamps (profiles/step_stamps.py).
#ifdef VNET_STAMPS
static __device__ long long* g_stamps = nullptr;
#define VNET_STAMP_DECL long long ts[12]; bool stamp_on = false
#define VNET_STAMP_STEP(step) stamp_on = blockIdx.x == 88 && (step) >= 4 && (step) < 8
#define VNET_STAMP(k) do { if (stamp_on) ts[k] = __builtin_readcyclecounter(); } while (0)
#define VNET_STAMP_FLUSH(step, wave, lane, n) do { if (stamp_on && (lane) == 0 && g_stamps) { \
        for (int q_ = 0; q_ < (n); ++q_) g_stamps[(((step) - 4) * 8 + (wave)) * 12 + q_] = ts[q_]; } } while (0)
#else
#define VNET_STAMP_DECL
#define VNET_STAMP_STEP(step)
#define VNET_STAMP(k) do {} while (0)
#define VNET_STAMP_FLUSH(step, wave, lane, n) do {} while (0)
#endif

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), every lane gets the sum: four v_add_f32_dpp (quad_perm [1,0,3,2],
// [2,3,0,1], row_half_mirror, row_mirror).  __shfl_xor compiles to ds_bpermute_b32, which goes through the LDS pipe and queues
// behind the other waves' fragment reads: the 16 dependent rounds of the statistics reduction were ~2.5 K cycles of the 16-cout
// kernel's brick step (s_memtime stamps, round 3).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}
// ... over 32 lanes (lanes 32h .. 32h+31): the row sums, then one exchange between the two rows
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
}

// cross-wave stage of the epilogue statistics: per-wave sums in red[wave][2 * CW] -> one row of the partial buffer
template <int WAVES_, int CW>
__device__ __forceinline__ void stats_row_write(const float* red, float* __restrict__ stats, size_t row, int co0, int Cout, int tid) {
    if (tid < 2 * CW) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_; ++w) t += red[w * 2 * CW + tid];
        const int a2 = tid / CW, c = co0 + (tid - a2 * CW);
        if (c < Cout) stats[row * 2 * Cout + (size_t)a2 * Cout + c] = t;
    }
}

// bf16-storage mode (the bf16 5^3 kernels and the IO16 instantiations): y0 / y1 / accsrc / res of ConvArgs then point at bf16 tensors (same NDHWC
// indexing, 2-byte elements).  One lane's 4 consecutive output channels `co..co+3` of voxel `ov` (bias already added):
// optional accumulation onto the stored gradient, ONE rounding (RNE), statistics of the ROUNDED values (+ residual) -- the
// batch-norm behind normalises exactly the tensor that is stored -- and an 8-byte store.
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
template <bool STATS>
__device__ __forceinline__ void epilogue4_b16(const ConvArgs& a, size_t ov, int co, float (&e)[4], float* s1, float* s2) {
    unsigned short* y = (co < a.Cy0) ? reinterpret_cast<unsigned short*>(a.y0) + ov * a.Cy0 + co
                                     : reinterpret_cast<unsigned short*>(a.y1) + ov * a.Cy1 + (co - a.Cy0);
    if (a.accum) {
        const unsigned short* src = a.accsrc ? reinterpret_cast<const unsigned short*>(a.accsrc) + ov * a.Cy0 + co : y;
        const uint2 o = *reinterpret_cast<const uint2*>(src);
        e[0] += bf_lo(o.x); e[1] += bf_hi(o.x); e[2] += bf_lo(o.y); e[3] += bf_hi(o.y);
    }
    const uint2 pk = make_uint2(pk_bf16(e[0], e[1]), pk_bf16(e[2], e[3]));
    if constexpr (STATS) {
        float v[4] = {bf_lo(pk.x), bf_hi(pk.x), bf_lo(pk.y), bf_hi(pk.y)};
        if (a.res) {
            const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.res) + ov * a.Cout + co);
            v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s1[k] += v[k]; s2[k] += v[k] * v[k]; }
    }
    *reinterpret_cast<uint2*>(y) = pk;
}

// The same for N units of one lane at once (round 3): every load of the batch -- the other gradient in accumulate mode, the
// residual of the statistics -- is in flight before the first is used.  One unit at a time, each runtime `if` around a load costs
// its own s_waitcnt vmcnt(0): s_memtime stamps showed 2.3 K (plain) to 5 K cycles (statistics + residual) per brick step of the
// persistent kernels in an epilogue that stores 32 bytes per lane.  Units that must not be stored (ok = false) come with ov = 0 and
// co = 0, a readable address.  e: in = accumulator + bias, out = what the batch-norm statistics see (rounded value + residual).
template <bool STATS, int N>
__device__ __forceinline__ void epilogue_b16_batch(const ConvArgs& a, const size_t (&ov)[N], const int (&co)[N], const bool (&ok)[N],
                                                   float (&e)[N][4]) {
    unsigned short* y[N];
    uint2 old[N], rr[N];
#pragma unroll
    for (int u = 0; u < N; ++u)
        y[u] = (co[u] < a.Cy0) ? reinterpret_cast<unsigned short*>(a.y0) + ov[u] * a.Cy0 + co[u]
                               : reinterpret_cast<unsigned short*>(a.y1) + ov[u] * a.Cy1 + (co[u] - a.Cy0);
    if (a.accum) {
        if (a.accsrc) {
#pragma unroll
            for (int u = 0; u < N; ++u)
                old[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.accsrc) + ov[u] * a.Cy0 + co[u]);
        } else {
#pragma unroll
            for (int u = 0; u < N; ++u) old[u] = *reinterpret_cast<const uint2*>(y[u]);
        }
    }
    if constexpr (STATS) {
        if (a.res) {
#pragma unroll
            for (int u = 0; u < N; ++u)
                rr[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.res) + ov[u] * a.Cout + co[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < N; ++u) {
        if (a.accum) { e[u][0] += bf_lo(old[u].x); e[u][1] += bf_hi(old[u].x); e[u][2] += bf_lo(old[u].y); e[u][3] += bf_hi(old[u].y); }
        const uint2 pk = make_uint2(pk_bf16(e[u][0], e[u][1]), pk_bf16(e[u][2], e[u][3]));
        if constexpr (STATS) {
            e[u][0] = bf_lo(pk.x); e[u][1] = bf_hi(pk.x); e[u][2] = bf_lo(pk.y); e[u][3] = bf_hi(pk.y);
            if (a.res) { e[u][0] += bf_lo(rr[u].x); e[u][1] += bf_hi(rr[u].x); e[u][2] += bf_lo(rr[u].y); e[u][3] += bf_hi(rr[u].y); }
        }
        if (ok[u]) *reinterpret_cast<uint2*>(y[u]) = pk;
    }
}

template <int KS, int STRIDE, int TZ, int TY, int TX, int KX = KS>
struct TileGeom {
    static constexpr int IZ = (TZ - 1) * STRIDE + KS;
    static constexpr int IY = (TY - 1) * STRIDE + KS;
    static constexpr int IX = (TX - 1) * STRIDE + KX;     // KX: kernel extent along x (1 for the x-im2col'ed input conv)
    static constexpr int NVOX_IN = IZ * IY * IX;
    static constexpr int LDS_FLOATS = NVOX_IN * 16;
};

// Tile staging, split in two halves so a tile's global loads can fly while MFMAs run:
//   issue  : every thread puts ALL of its 16-byte loads in flight into registers (unconditional loads from
//            clamped addresses + select: a branch around a load would make hipcc wait vmcnt(0) per element);
//   commit : registers -> LDS image [iz][iy][ix][16 channels].
// Each thread owns one (x, channel-quad) column of the tile and walks the (z,y) rows, so the per-load address
// arithmetic is a handful of integer ops (a flat quad index would cost ~60 VALU per load in div/mod + 64-bit mads).
// S16: the source tensors are bf16 (bf16-storage mode of the 2^3 convolutions): 8-byte loads, converted to fp32 on commit.
template <int IZ, int IY, int IX, int NT, bool S16 = false>
struct XTile {
    using V = typename std::conditional<S16, uint2, float4>::type;          // one (voxel, channel quad) unit in registers
    using E = typename std::conditional<S16, unsigned short, float>::type;  // source element
    static __device__ __forceinline__ V vzero() { V z; if constexpr (S16) { z.x = 0u; z.y = 0u; } else { z.x = 0.f; z.y = 0.f; z.z = 0.f; z.w = 0.f; } return z; }
    static __device__ __forceinline__ float4 to_f4(const V& v) {
        if constexpr (S16) return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                              __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
        else return v;
    }
    static constexpr int COLS = IX * 4;
    static constexpr int RPI = NT / COLS;
    static constexpr int ROWS = IZ * IY;
    static constexpr int PER = (ROWS + RPI - 1) / RPI;
    static_assert(RPI >= 1, "tile row wider than the workgroup");
    // Address arithmetic is the VALU cost of staging (every non-MFMA VALU instruction takes MFMA issue time): one 64-bit
    // per-thread base (source tensor, batch, x, channel) and a 32-bit row offset gz*Hi*Wi*Cs + gy*Wi*Cs per load (one
    // sample's volume x channels stays below 2^31 elements), rows stepped incrementally (RPI rows per load).
    template <int K0, int KN>
    __device__ static __forceinline__ void issue_part(V (&v)[KN], const float* __restrict__ x0f, const float* __restrict__ x1f,
                                                      int C0, int C1, int chunk, int b, int gz0, int gy0, int gx0,
                                                      int Di, int Hi, int Wi, int tid) {
        const E* __restrict__ x0 = reinterpret_cast<const E*>(x0f);
        const E* __restrict__ x1 = reinterpret_cast<const E*>(x1f);
        const int r0 = tid / COLS, col = tid - r0 * COLS;
        const int ix = col >> 2, cq = col & 3;
        const int c = chunk * 16 + cq * 4;
        const int gx = gx0 + ix;
        const bool colok = r0 < RPI && (unsigned)gx < (unsigned)Wi && c < C0 + C1;
        const bool first = c < C0 || !colok;
        const int Cs = first ? C0 : C1;
        const int rowstride = Wi * Cs, planestride = Hi * rowstride;
        const E* bp = (first ? x0 + (colok ? c : 0) : x1 + (c - C0)) + (size_t)b * Di * planestride + (colok ? gx * Cs : 0);
        constexpr int DIZ = RPI / IY, DIY = RPI % IY;
        int row = r0 + K0 * RPI;
        int iz = row / IY, iy = row - iz * IY;
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            const int gz = gz0 + iz, gy = gy0 + iy;
            const bool ok = colok && row < ROWS && (unsigned)gz < (unsigned)Di && (unsigned)gy < (unsigned)Hi;
            const int off = ok ? gz * planestride + gy * rowstride : 0;
            const V t = *reinterpret_cast<const V*>(bp + off);
            v[k] = ok ? t : vzero();
            row += RPI; iy += DIY; iz += DIZ;
            if (iy >= IY) { iy -= IY; ++iz; }
        }
    }
    __device__ static __forceinline__ void issue(V (&v)[PER], const float* __restrict__ x0, const float* __restrict__ x1,
                                                 int C0, int C1, int chunk, int b, int gz0, int gy0, int gx0,
                                                 int Di, int Hi, int Wi, int tid) {
        issue_part<0, PER>(v, x0, x1, C0, C1, chunk, b, gz0, gy0, gx0, Di, Hi, Wi, tid);
    }
    __device__ static __forceinline__ void commit(float* lds, const V (&v)[PER], int tid) {
        const int r0 = tid / COLS, col = tid - r0 * COLS;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int row = r0 + k * RPI;
            if (r0 < RPI && row < ROWS) *reinterpret_cast<float4*>(lds + ((size_t)row * IX) * 16 + col * 4) = to_f4(v[k]);
        }
    }
};

// Synchronous staging of one 16-channel chunk of the input brick (+halo) into LDS as [iz][iy][ix][16].
template <int IZ, int IY, int IX, int NT, bool S16 = false>
__device__ __forceinline__ void load_tile(float* lds, const float* __restrict__ x0, const float* __restrict__ x1,
                                          int C0, int C1, int vec_in, int chunk, int b, int gz0, int gy0, int gx0,
                                          int Di, int Hi, int Wi, int tid) {
    constexpr int NQ = IZ * IY * IX * 4;
    const int Cin = C0 + C1;
    if (S16 || vec_in) {                                   // (bf16 sources: the entry points insist on channel multiples of 4)
        using XT = XTile<IZ, IY, IX, NT, S16>;
        typename XT::V v[XT::PER];
        XT::issue(v, x0, x1, C0, C1, chunk, b, gz0, gy0, gx0, Di, Hi, Wi, tid);
        XT::commit(lds, v, tid);
        return;
    }
    for (int q = tid; q < NQ; q += NT) {      // channel counts that are not multiples of 4: scalar gather
        const int vox = q >> 2, cq = q & 3;
        const int ix = vox % IX, iy = (vox / IX) % IY, iz = vox / (IX * IY);
        const int gz = gz0 + iz, gy = gy0 + iy, gx = gx0 + ix;
        const int c = chunk * 16 + cq * 4;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gz < (unsigned)Di && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi && c < Cin) {
            const size_t gv = ((size_t)(b * Di + gz) * Hi + gy) * Wi + gx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ck = c + k;
                if (ck < Cin) e[k] = (ck < C0) ? x0[gv * C0 + ck] : x1[gv * C1 + (ck - C0)];
            }
        }
        *reinterpret_cast<float4*>(lds + (size_t)q * 4) = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// STATS: separate instantiation with the batch-norm statistics in the epilogue -- kept out of the plain kernels, whose main
// loop lost 2-3 % when the (unused) epilogue code was merely present (register allocation / code placement)
// IO16: bf16-storage mode of the 2^3 down / transposed convolutions -- bf16 tensors in (converted to fp32 while the tile is
// committed to LDS; the packed filter holds bf16-rounded values, so every product is the exact bf16 x bf16 product, accumulated
// in fp32 like the matrix cores' bf16 path) and bf16 tensors out.
template <int KS, int STRIDE, int TZ, int TY, int TX, int WAVES, int MS, int NS, bool UP, int KX = KS, bool STATS = false, bool IO16 = false>
__global__ void __launch_bounds__(WAVES * 64) conv_kernel(ConvArgs a) {
    using G = TileGeom<KS, STRIDE, TZ, TY, TX, KX>;
    static_assert(TZ * TY * TX == WAVES * MS * 16, "brick must be WAVES*MS 16-voxel subtiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    // (static priority for the second half of the waves, which helps the 8-wave filter-gradient kernel, measured +0.27 ms
    // per step here: two 4-wave workgroups share a CU and the priority then favours one workgroup's waves over the other's)

    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    int brick = xcd_remap(blockIdx.x, nbrick);
    const int brick_id = brick;
    const int bx = brick % a.nbx; brick /= a.nbx;
    const int by = brick % a.nby; brick /= a.nby;
    const int bz = brick % a.nbz; const int b = brick / a.nbz;
    const int co0 = blockIdx.y * (NS * 16);
    const int split = blockIdx.z / a.nz, zsplit = blockIdx.z - split * a.nz;
    const int dz0 = zsplit * KS / a.nz, dz1 = (zsplit + 1) * KS / a.nz;
    const int c_begin = split * a.cps;
    const int c_end = min(a.nchunks, c_begin + a.cps);

    // per-lane LDS offsets of this wave's MS voxel subtiles (B operand: voxel = lane&15, k-group = lane>>4)
    int boff[MS];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int v = (wave * MS + m) * 16 + i;
        const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
        boff[m] = ((vz * STRIDE * G::IY + vy * STRIDE) * G::IX + vx * STRIDE) * 16 + kk * 4;
    }

    f32x4 acc[MS][NS];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int gz0 = bz * TZ * STRIDE - a.pad, gy0 = by * TY * STRIDE - a.pad, gx0 = bx * TX * STRIDE - a.padx;
    const size_t tap_stride = (size_t)a.CQ * a.CoutP;   // float4 units

    // 2^3 / transposed kernels: a brick has only 8 (1) taps of MFMA work per chunk, so an L2 round trip per tap in front of
    // its MFMAs would dominate.  The first RG taps' weight fragments are issued BEFORE the tile staging (in flight together
    // with the tile's HBM loads), the rest stream through the same RG-slot ring RG taps ahead.
    constexpr int T3G = KS * KS * KS, RG = (KS == 5) ? 1 : (T3G < 4 ? T3G : 4);
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const float4* wq = a.wp + ((size_t)(chunk * 4 + kk) * a.CoutP + co0 + i);
        float4 wg[RG][NS];
        if constexpr (KS == 2) {
#pragma unroll
            for (int j = 0; j < RG; ++j)
#pragma unroll
                for (int n = 0; n < NS; ++n) wg[j][n] = wq[(size_t)j * tap_stride + n * 16];
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        load_tile<G::IZ, G::IY, G::IX, WAVES * 64, IO16>(lds, a.x0, a.x1, a.C0, a.C1, a.vec_in, chunk, b, gz0, gy0, gx0,
                                                         a.Di, a.Hi, a.Wi, tid);
        __syncthreads();
        if constexpr (KS == 1) {      // transposed conv: one tap per chunk; its fragment is fetched here, together with the LDS reads
                                      // (issuing it ahead of the staging or of the barrier measured 18-25 % slower)
#pragma unroll
            for (int n = 0; n < NS; ++n) wg[0][n] = wq[n * 16];
        }
        if constexpr (KS == 5) {
            // Weight fragments come straight from L2 (the whole filter is shared by every workgroup), so
            // they are software-pipelined PF taps ahead through a 5-slot register ring (5 | 25 taps per
            // dz slab keeps every ring index a compile-time constant) -- the L2 round trip hides under
            // the MFMAs of the taps in between instead of stalling each tap.
            constexpr int T2 = KS * KX, T3 = T2 * KS, R = 5, PF = (NS == 1) ? 3 : 1;
            static_assert(T2 % R == 0, "ring slots must tile a dz slab");
            float4 wf[R][NS];
            float4 xf[R][MS];     // B fragments ride the same ring one tap ahead (LDS latency off the MFMA path)
#pragma unroll
            for (int j = 0; j < PF; ++j)
#pragma unroll
                for (int n = 0; n < NS; ++n) wf[j][n] = wq[(size_t)(dz0 * T2 + j) * tap_stride + n * 16];
#pragma unroll
            for (int m = 0; m < MS; ++m) xf[0][m] = *reinterpret_cast<const float4*>(lds + dz0 * G::IY * G::IX * 16 + boff[m]);
            for (int dz = dz0; dz < dz1; ++dz) {
#pragma unroll
                for (int t2 = 0; t2 < T2; ++t2) {
                    {
                        const int tn = min(dz * T2 + t2 + PF, T3 - 1);
                        const float4* wn = wq + (size_t)tn * tap_stride;
#pragma unroll
                        for (int n = 0; n < NS; ++n) wf[(t2 + PF) % R][n] = wn[n * 16];
                        const int t2n = (t2 + 1) % T2;
                        const int dzn = min(dz + (t2 == T2 - 1 ? 1 : 0), KS - 1);
                        const float* ln = lds + ((dzn * G::IY + t2n / KX) * G::IX + t2n % KX) * 16;
#pragma unroll
                        for (int m = 0; m < MS; ++m) xf[(t2 + 1) % R][m] = *reinterpret_cast<const float4*>(ln + boff[m]);
                        // pin the issue point: hipcc otherwise sinks the loads to just before their first use
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const float4* wc = wf[t2 % R];
                    const float4* xc = xf[t2 % R];
                    // element j of both fragments feeds MFMA step j; consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int m = 0; m < MS; ++m)
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[n].x, xc[m].x, acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MS; ++m)
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[n].y, xc[m].y, acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MS; ++m)
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[n].z, xc[m].z, acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MS; ++m)
#pragma unroll
                        for (int n = 0; n < NS; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[n].w, xc[m].w, acc[m][n], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < T3G; ++t) {
                const int dz = t / (KS * KS), dy = (t / KS) % KS, dx = t % KS;
                const float* lrow = lds + ((dz * G::IY + dy) * G::IX + dx) * 16;
                float4 xf[MS];
#pragma unroll
                for (int m = 0; m < MS; ++m) xf[m] = *reinterpret_cast<const float4*>(lrow + boff[m]);
                const float4* wf = wg[t % RG];
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int n = 0; n < NS; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n].x, xf[m].x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n].y, xf[m].y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n].z, xf[m].z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n].w, xf[m].w, acc[m][n], 0, 0, 0);
                    }
                if (t + RG < T3G) {
#pragma unroll
                    for (int n = 0; n < NS; ++n) wg[t % RG][n] = wq[(size_t)(t + RG) * tap_stride + n * 16];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // transposed conv whose O is no multiple of 4: a lane's four values can straddle two taps -> a separate, scalar
    // epilogue (kept out of the vector epilogue's loops: inlined there it cost the O % 4 == 0 kernels 10-18 %)
    if (!IO16 && UP && (a.upO & 3)) {
#pragma unroll
        for (int m = 0; m < MS; ++m) {      // (static indices: a dynamic index would push acc[][] to scratch for the whole kernel)
            const int v = (wave * MS + m) * 16 + i;
            const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
            const int oz = bz * TZ + vz, oy = by * TY + vy, ox = bx * TX + vx;
            if (oz >= a.Di || oy >= a.Hi || ox >= a.Wi) continue;
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const int cop = co0 + n * 16 + kk * 4;
                const float e[4] = {acc[m][n].x, acc[m][n].y, acc[m][n].z, acc[m][n].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ck = cop + k;
                    if (ck >= 8 * a.upO) break;
                    const int tp = ck / a.upO, o1 = ck - tp * a.upO;
                    const int z1 = 2 * oz + (tp >> 2), y1 = 2 * oy + ((tp >> 1) & 1), x1 = 2 * ox + (tp & 1);
                    if (z1 >= a.Do || y1 >= a.Ho || x1 >= a.Wo) continue;
                    float* dst = a.y0 + (((size_t)(b * a.Do + z1) * a.Ho + y1) * a.Wo + x1) * a.upO + o1;
                    *dst = e[k] + (a.bias ? a.bias[o1] : 0.f) + (a.accum ? *dst : 0.f);
                }
            }
        }
        return;
    }

    // epilogue: lane holds cout = co0 + n*16 + 4*kk + {0..3} of voxel (m, i)
    constexpr int NSS = STATS ? NS : 1;
    float s1[NSS][4], s2[NSS][4];        // batch-norm statistics of this lane's channels (STATS)
#pragma unroll
    for (int n = 0; n < NSS; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) s1[n][k] = s2[n][k] = 0.f;
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int v = (wave * MS + m) * 16 + i;
        const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
        const int oz = bz * TZ + vz, oy = by * TY + vy, ox = bx * TX + vx;
        if (UP) {
            if (oz >= a.Di || oy >= a.Hi || ox >= a.Wi) continue;
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const int cop = co0 + n * 16 + kk * 4;          // index into N' = 8*O
                if (cop >= 8 * a.upO) continue;
                const int tap = cop / a.upO, o = cop - tap * a.upO;
                const int zz = 2 * oz + (tap >> 2), yy = 2 * oy + ((tap >> 1) & 1), xx = 2 * ox + (tap & 1);
                if (zz >= a.Do || yy >= a.Ho || xx >= a.Wo) continue;
                const size_t ov = ((size_t)(b * a.Do + zz) * a.Ho + yy) * a.Wo + xx;
                f32x4 r = acc[m][n];
                if (a.bias) { r.x += a.bias[o]; r.y += a.bias[o + 1]; r.z += a.bias[o + 2]; r.w += a.bias[o + 3]; }
                if constexpr (IO16) {
                    unsigned short* d16 = reinterpret_cast<unsigned short*>(a.y0) + ov * a.upO + o;
                    if (a.accum) {
                        const uint2 old = *reinterpret_cast<const uint2*>(d16);
                        r.x += bf_lo(old.x); r.y += bf_hi(old.x); r.z += bf_lo(old.y); r.w += bf_hi(old.y);
                    }
                    *reinterpret_cast<uint2*>(d16) = make_uint2(pk_bf16(r.x, r.y), pk_bf16(r.z, r.w));
                    continue;
                }
                float4* dst = reinterpret_cast<float4*>(a.y0 + ov * a.upO + o);
                if (a.accum) { const float4 old = *dst; r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
                *dst = make_float4(r.x, r.y, r.z, r.w);
            }
        } else {
            if (oz >= a.Do || oy >= a.Ho || ox >= a.Wo) continue;
            const size_t ov = ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                const int co = co0 + n * 16 + kk * 4;
                f32x4 r = acc[m][n];
                if (a.part) {
                    *reinterpret_cast<float4*>(a.part + blockIdx.z * a.part_stride + ov * a.CoutP + co) =
                        make_float4(r.x, r.y, r.z, r.w);
                    continue;
                }
                if (co >= a.Cout) continue;
                float e[4] = {r.x, r.y, r.z, r.w};
                if constexpr (IO16) {
                    if (a.bias) { e[0] += a.bias[co]; e[1] += a.bias[co + 1]; e[2] += a.bias[co + 2]; e[3] += a.bias[co + 3]; }
                    epilogue4_b16<STATS>(a, ov, co, e, s1[STATS ? n : 0], s2[STATS ? n : 0]);
                    continue;
                }
                if (a.vec_out && co + 3 < a.Cout) {
                    if (a.bias) { e[0] += a.bias[co]; e[1] += a.bias[co + 1]; e[2] += a.bias[co + 2]; e[3] += a.bias[co + 3]; }
                    if constexpr (STATS) {
                        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (a.res) rr = *reinterpret_cast<const float4*>(a.res + ov * a.Cout + co);
                        const float vv[4] = {e[0] + rr.x, e[1] + rr.y, e[2] + rr.z, e[3] + rr.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) { s1[n][k] += vv[k]; s2[n][k] += vv[k] * vv[k]; }
                    }
                    float* p = (co < a.Cy0) ? a.y0 + ov * a.Cy0 + co : a.y1 + ov * a.Cy1 + (co - a.Cy0);
                    if (a.accum) { const float4 old = *reinterpret_cast<const float4*>(p); e[0] += old.x; e[1] += old.y; e[2] += old.z; e[3] += old.w; }
                    *reinterpret_cast<float4*>(p) = make_float4(e[0], e[1], e[2], e[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ck = co + k;
                        if (ck >= a.Cout) break;
                        float* dst = (ck < a.Cy0) ? a.y0 + ov * a.Cy0 + ck : a.y1 + ov * a.Cy1 + (ck - a.Cy0);
                        *dst = e[k] + (a.bias ? a.bias[ck] : 0.f) + (a.accum ? *dst : 0.f);
                    }
                }
            }
        }
    }
    if constexpr (STATS && !UP) if (!a.part) {
        // lanes of one kk group hold the same channels for 16 different voxels: butterfly over i, then across waves via LDS
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    s1[n][k] += __shfl_xor(s1[n][k], off, 64);
                    s2[n][k] += __shfl_xor(s2[n][k], off, 64);
                }
        constexpr int CW = NS * 16;
        __syncthreads();                                   // every wave is done with the LDS tile
        if (i == 0) {
#pragma unroll
            for (int n = 0; n < NS; ++n)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lds[wave * 2 * CW + n * 16 + kk * 4 + k] = s1[n][k];
                    lds[wave * 2 * CW + CW + n * 16 + kk * 4 + k] = s2[n][k];
                }
        }
        __syncthreads();
        stats_row_write<WAVES, CW>(lds, a.stats, (size_t)brick_id, co0, a.Cout, tid);
    }
}

// bf16-storage twin of the split-K reduce below: y0 / y1 / res / accsrc are bf16 tensors, statistics of the rounded values
__global__ void splitk_reduce_b16_kernel(const float* __restrict__ part, size_t part_stride, int nsplit,
                                         const float* __restrict__ bias, unsigned short* y0, unsigned short* y1, int Cy0, int Cy1,
                                         int CoutP, size_t nvox, int accum, const unsigned short* __restrict__ res, float* __restrict__ stats,
                                         const unsigned short* accsrc) {
    const int Cout = Cy0 + Cy1;
    const size_t total = nvox * (size_t)Cout;
    float t1 = 0.f, t2 = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t v = idx / Cout; const int c = (int)(idx - v * Cout);
        float s = bias ? bias[c] : 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[k * part_stride + v * CoutP + c];
        unsigned short* dst = (c < Cy0) ? y0 + v * Cy0 + c : y1 + v * Cy1 + (c - Cy0);
        if (accum) s += __uint_as_float((uint32_t)(accsrc ? accsrc[idx] : *dst) << 16);
        const uint32_t h = pk_bf16(s, 0.f) & 0xffffu;
        if (stats) { const float w = __uint_as_float(h << 16) + (res ? __uint_as_float((uint32_t)res[idx] << 16) : 0.f); t1 += w; t2 += w * w; }
        *dst = (unsigned short)h;
    }
    if (stats) {
        __shared__ float sh[2][256];
        sh[0][threadIdx.x] = t1; sh[1][threadIdx.x] = t2;
        __syncthreads();
        for (int q = threadIdx.x; q < 2 * Cout; q += 256) {
            const int a2 = q / Cout, c = q - a2 * Cout;
            float t = 0.f;
            for (int k = c; k < 256; k += Cout) t += sh[a2][k];
            stats[(size_t)blockIdx.x * 2 * Cout + (size_t)a2 * Cout + c] = t;
        }
    }
}

// y = sum_s part[s] + bias, scattered to the (possibly dual) NDHWC destination
__global__ void splitk_reduce_kernel(const float* __restrict__ part, size_t part_stride, int nsplit,
                                     const float* __restrict__ bias, float* y0, float* y1, int Cy0, int Cy1,
                                     int CoutP, size_t nvox, int accum, const float* __restrict__ res, float* __restrict__ stats,
                                     const float* accsrc = nullptr) {
    const int Cout = Cy0 + Cy1;
    const size_t total = nvox * (size_t)Cout;
    // statistics (stats != NULL): 256 % Cout == 0, so a thread meets ONE channel (tid % Cout) on its whole grid-stride walk
    float t1 = 0.f, t2 = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t v = idx / Cout; const int c = (int)(idx - v * Cout);
        float s = bias ? bias[c] : 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[k * part_stride + v * CoutP + c];
        if (stats) { const float w = s + (res ? res[idx] : 0.f); t1 += w; t2 += w * w; }
        float* dst = (c < Cy0) ? y0 + v * Cy0 + c : y1 + v * Cy1 + (c - Cy0);
        *dst = accum ? (accsrc ? accsrc[idx] : *dst) + s : s;      // accsrc: single output (Cy1 == 0), same layout as y0
    }
    if (stats) {
        __shared__ float sh[2][256];
        sh[0][threadIdx.x] = t1; sh[1][threadIdx.x] = t2;
        __syncthreads();
        for (int q = threadIdx.x; q < 2 * Cout; q += 256) {
            const int a2 = q / Cout, c = q - a2 * Cout;
            float t = 0.f;
            for (int k = c; k < 256; k += Cout) t += sh[a2][k];
            stats[(size_t)blockIdx.x * 2 * Cout + (size_t)a2 * Cout + c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// filter gradient:  D[row = cout][col = cin] += A[cout][k = voxel] * B[voxel][cin(tap-shifted)]
// A workgroup = 4 waves; wave w owns TW taps (all of the workgroup's NS*16 cout x 16 cin), so the
// x tile in LDS is re-used by 4*TW taps and the dy tile by all of them.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x0; const float* x1; int C0, C1, Cin;
    const float* dy; int Cout;
    int B, Di, Hi, Wi, Do, Ho, Wo;
    int CinP, CoutP, ncob;
    int nbz, nby, nbx, nbrick, nsplit;
    int pad, padx, vec_in, vec_dy;
    float* part;   // [split][tap][CinP][CoutP]
};

// IO16: x and dy are bf16 tensors (bf16-storage mode of the 2^3 convolutions), converted to fp32 on their way into LDS
template <int KS, int STRIDE, int TZ, int TY, int TX, int NS, int TW, int KX = KS, bool IO16 = false>
__global__ void __launch_bounds__(512) wgrad_kernel(WgradArgs a) {
#define VNET_WG_BID_X blockIdx.x
#define VNET_WG_BID_Y blockIdx.y
#define VNET_WG_BID_Z blockIdx.z
#include "wgrad_body.inc"
#undef VNET_WG_BID_X
#undef VNET_WG_BID_Y
#undef VNET_WG_BID_Z
}

// the same body for a kernel that decodes (split, block, tap group) itself (conv_b16.hip: the grouped launch)
template <int KS, int STRIDE, int TZ, int TY, int TX, int NS, int TW, int KX = KS, bool IO16 = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int gid_x, const int gid_y, const int gid_z) {
#define VNET_WG_BID_X gid_x
#define VNET_WG_BID_Y gid_y
#define VNET_WG_BID_Z gid_z
#include "wgrad_body.inc"
#undef VNET_WG_BID_X
#undef VNET_WG_BID_Y
#undef VNET_WG_BID_Z
}

// dw = sum over splits of the partial filter gradients; 64 output groups x 4 split-lanes per workgroup.  VEC: a group is
// 4 consecutive cout (16-byte accesses; needs Cout % 4 == 0), otherwise one output.  Four independent partial sums per
// lane: the slab loads of one output are 256 KB or more apart, so a serial chain would wait one HBM round trip per slab
// (the summation order is fixed -> deterministic).
// blk / nblk: this workgroup's index and the number of workgroups that share the job (the whole grid, or one job's slice of a batch)
template <bool VEC>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, int nsplit, int T3, int CinP, int CoutP,
                                                  int Cin, int Cout, float* __restrict__ dw, unsigned blk, unsigned nblk) {
    constexpr int W = VEC ? 4 : 1;
    __shared__ float sh[4][64 * W];
    const size_t total = (size_t)T3 * Cin * Cout / W;               // output groups
    const size_t sstride = (size_t)T3 * CinP * CoutP;
    const int CoutG = Cout / W;
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    for (size_t base = (size_t)blk * 64; base < total; base += (size_t)nblk * 64) {
        const size_t idx = base + o;
        float s[W];
#pragma unroll
        for (int j = 0; j < W; ++j) s[j] = 0.f;
        if (idx < total) {
            const int co = (int)(idx % CoutG) * W;
            const size_t r = idx / CoutG;
            const int ci = (int)(r % Cin), t = (int)(r / Cin);
            const float* p = part + ((size_t)t * CinP + ci) * CoutP + co;
            float acc[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < W; ++j) acc[u][j] = 0.f;
            auto ld = [&](int k, float (&d)[W]) {
                if (VEC) { const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * sstride); d[0] += v.x; d[W > 1 ? 1 : 0] += v.y; d[W > 2 ? 2 : 0] += v.z; d[W > 3 ? 3 : 0] += v.w; }
                else d[0] += p[(size_t)k * sstride];
            };
            int k = sg;
            for (; k + 12 < nsplit; k += 16) { ld(k, acc[0]); ld(k + 4, acc[1]); ld(k + 8, acc[2]); ld(k + 12, acc[3]); }
            for (; k < nsplit; k += 4) ld(k, acc[0]);
#pragma unroll
            for (int j = 0; j < W; ++j) s[j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < W; ++j) sh[sg][o * W + j] = s[j];
        __syncthreads();
        if (sg == 0 && idx < total) {
            float out[W];
#pragma unroll
            for (int j = 0; j < W; ++j) out[j] = sh[0][o * W + j] + sh[1][o * W + j] + sh[2][o * W + j] + sh[3][o * W + j];
            if (VEC) *reinterpret_cast<float4*>(dw + idx * 4) = make_float4(out[0], out[W > 1 ? 1 : 0], out[W > 2 ? 2 : 0], out[W > 3 ? 3 : 0]);
            else dw[idx] = out[0];
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int T3, int CinP, int CoutP,
                                                           int Cin, int Cout, float* __restrict__ dw) {
    wgrad_reduce_body<VEC>(part, nsplit, T3, CinP, CoutP, Cin, Cout, dw, blockIdx.x, gridDim.x);
}

// Deferred reduces (round 2): with vnet_wgrad_defer(1) the filter-gradient entry points leave their partial slabs in the caller's
// workspace and queue the reduce; vnet_wgrad_flush() runs every queued reduce in ONE launch (26 launches of ~7 us per V-Net
// step otherwise).  The job table travels by value in the kernel arguments (a captured graph keeps it), results are bit-identical
// to the per-layer reduce (same summation order).
}  // namespace
namespace vnet_detail {
struct ReduceJob { const float* part; float* dw; int nsplit, T3, CinP, CoutP, Cin, Cout, vec; unsigned blk0; };
constexpr int REDUCE_BATCH = 32;
struct ReduceBatch { ReduceJob job[REDUCE_BATCH]; int n; };
// one queue per process, whichever translation unit launches the filter-gradient kernels (defined in conv_mfma.hip)
// One queue PER STREAM (round 5): a filter-gradient entry point defers iff deferral is on for the stream it launches on, and a
// flush takes that stream's jobs only -- models that work on different streams (or host threads) never see each other's queue.
// (Rounds 2-4: one process-wide queue; a flush on stream A would have run the reduces queued from stream B.)
struct DeferQueue { bool on = false; std::vector<ReduceJob> pending; };
struct DeferState { std::mutex mu; std::unordered_map<hipStream_t, DeferQueue> q; };
__attribute__((visibility("hidden"))) DeferState& defer_state();
// Tuning switches of the library: read ONCE from the environment (first use), changed afterwards only through vnet_set_option --
// no getenv on a launch path (round 5; rounds 2-4 read seven variables per call).  One instance per process (conv_mfma.hip).
struct Tuning {
    int wgrad_zs;            // VNET_WGRAD_ZS: z-streaming filter gradient (bf16): 0 never, 1 wherever it applies, 2 (default) rows < 32 voxels, grouped launch only
    int wgrad_rr;            // VNET_WGRAD_RR: row-reuse filter gradient (bf16): 0 never, 1 (default) where it pays, 2 wherever it applies
    int conv_in4;            // VNET_CONV_IN4: x-im2col form of the zero-padded network input (bf16 storage), default 1
    double group_rounds;     // VNET_WGRAD_GROUP_ROUNDS: rounds of the 256 CUs the grouped filter-gradient launch plans for (default 2; <= 0: every layer on its own)
    int group_debug;         // VNET_WGRAD_GROUP_DEBUG: print the group's plan
    int bf16_deep;           // VNET_BF16_DEEP: deep-level bf16 kernel (csrc/conv_deep.h), default 1
    int bf16_deep_target;    // VNET_BF16_DEEP_TARGET: workgroups its K split aims for (default 256)
    int bf16_c16pp;          // VNET_BF16_C16PP: ping-pong form of the 16-cout bf16 kernel (csrc/conv_c16pp.h) where it applies, default 1
    int x3_nb2;              // VNET_X3_NB2: f32x3 convolution, two 16-cout blocks per item where the layer allows (default 1)
    int f32_small;           // VNET_F32_SMALL: fp32 5^3 convolutions on volumes narrower than 16: 0 = 8x8x8 bricks / 8 waves (rounds 1-4),
                             // 1 = 4x8x8 bricks / 4 waves (two workgroups per CU), 2 (default) = ... and 4x4x4 bricks for volumes <= 4^3
                             // (profiles/r05_bench_small.txt: 4^3 x 2 256->256 135 -> 30 us, 8^3 x 2 128->128 65 -> 39 us, 8^3 256->256 71.6 -> 69.4 us)
};
__attribute__((visibility("hidden"))) Tuning& tuning();
}  // namespace vnet_detail
namespace {
using vnet_detail::ReduceJob; using vnet_detail::ReduceBatch; using vnet_detail::REDUCE_BATCH;
using vnet_detail::DeferState; using vnet_detail::DeferQueue; using vnet_detail::defer_state; using vnet_detail::tuning;

__global__ void __launch_bounds__(256) wgrad_reduce_batched_kernel(ReduceBatch b) {
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < b.n; ++k) if (blockIdx.x >= b.job[k].blk0) j = k;
    const ReduceJob& q = b.job[j];
    const unsigned nblk = (j + 1 < b.n ? b.job[j + 1].blk0 : gridDim.x) - q.blk0;
    if (q.vec) wgrad_reduce_body<true>(q.part, q.nsplit, q.T3, q.CinP, q.CoutP, q.Cin, q.Cout, q.dw, blockIdx.x - q.blk0, nblk);
    else wgrad_reduce_body<false>(q.part, q.nsplit, q.T3, q.CinP, q.CoutP, q.Cin, q.Cout, q.dw, blockIdx.x - q.blk0, nblk);
}


void launch_wgrad_reduce(const float* part, int nsplit, int T3, int CinP, int CoutP, int Cin, int Cout, float* dw, hipStream_t st) {
    const bool vec = (Cout % 4 == 0) && (CoutP % 4 == 0) && ((reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(part)) & 15) == 0;
    {
        DeferState& ds = defer_state();
        std::lock_guard<std::mutex> lk(ds.mu);
        auto it = ds.q.find(st);
        if (it != ds.q.end() && it->second.on) {
            it->second.pending.push_back(ReduceJob{part, dw, nsplit, T3, CinP, CoutP, Cin, Cout, vec ? 1 : 0, 0u});
            return;
        }
    }
    const size_t total = (size_t)T3 * Cin * Cout / (vec ? 4 : 1);
    const int blocks = (int)min((size_t)4096, (total + 63) / 64);
    if (vec) hipLaunchKernelGGL(wgrad_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, part, nsplit, T3, CinP, CoutP, Cin, Cout, dw);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, part, nsplit, T3, CinP, CoutP, Cin, Cout, dw);
}


// bf16 filter image: [cin chunk of 16][tap][cout block of 32][cin half][32 cout][8 cin]
__device__ __forceinline__ void pack_bf16_elem(int mode, const float* __restrict__ w, unsigned short* __restrict__ wp,
                                               int T, int I, int O, int ncob, size_t idx) {
    const int e = (int)(idx & 7);
    size_t q = idx >> 3;
    const int m = (int)(q & 31); q >>= 5;
    const int hf = (int)(q & 1); q >>= 1;
    const int cob = (int)(q % ncob); q /= ncob;
    const int t = (int)(q % T);
    const int chunk = (int)(q / T);
    const int k = chunk * 16 + hf * 8 + e, n = cob * 32 + m;
    float v = 0.f;
    if (mode == VNET_PACK_FWD_BF16) { if (k < I && n < O) v = w[((size_t)t * I + k) * O + n]; }
    else { if (k < O && n < I) v = w[((size_t)(T - 1 - t) * I + n) * O + k]; }
    wp[idx] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
}

#include "x3_pack.h"

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(int mode, const float* __restrict__ w, float* __restrict__ wp, int T, int I, int O,
                            int CQ, int NP, size_t total) {
    const bool r16 = (mode & VNET_PACK_ROUND_BF16) != 0;      // fp32 image of bf16-rounded values (bf16-storage mode of the 2^3 convolutions)
    mode &= ~VNET_PACK_ROUND_BF16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx & 3);
        size_t q = idx >> 2;
        const int n = (int)(q % NP); q /= NP;
        const int cq = (int)(q % CQ);
        const int t = (int)(q / CQ);
        const int k = cq * 4 + r;
        float v = 0.f;
        if (mode == VNET_PACK_FWD) {            // k = ci, n = co
            if (k < I && n < O) v = w[((size_t)t * I + k) * O + n];
        } else if (mode == VNET_PACK_BWD) {     // k = co_f, n = ci_f, flipped tap
            if (k < O && n < I) v = w[((size_t)(T - 1 - t) * I + n) * O + k];
        } else {                                // UP: w [8][O][I]; k = ci (I), n = a*O + o
            if (k < I && n < 8 * O) v = w[(size_t)n * I + k];
        }
        wp[idx] = r16 ? bf_lo(pk_bf16(v, 0.f)) : v;
    }
}

// s_setprio alternation between the two waves of a SIMD (w, w + 4), as in conv_x3.h: the matrix pipe goes to the OLDER wave whenever
// both are ready, so the older half of a workgroup reaches the step's common phases (stores, prefetch issue, barrier) early and the
// younger half finishes alone.  Round 5 (profiles/r05_ab_prio.txt, interleaved A/B on two boxes): 16-cout forward kernel -2.3 %
// (128^3 32->16: 0.285 -> 0.278 ms), row-reuse filter gradient -1.2 %; the row-pair kernel (a barrier per filter plane) did not
// move and has none; switching in the middle of the dz-pair phase instead of at its end: no gain.
#define VNET_PRIO_ALT(cond) do { if (cond) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } while (0)
#define VNET_PRIO_OFF() __builtin_amdgcn_s_setprio(0)

// all filters of a network in ONE launch (after every optimiser step): blockIdx.y selects the descriptor
// {w, wp, mode, T, I, O, CQ, NP} (8 x int64 in device memory)
__global__ void __launch_bounds__(256) pack_batched_kernel(const long long* __restrict__ descs) {
    const long long* d = descs + (size_t)blockIdx.y * 8;
    const float* w = reinterpret_cast<const float*>(d[0]);
    float* wp = reinterpret_cast<float*>(d[1]);
    const bool r16 = ((int)d[2] & VNET_PACK_ROUND_BF16) != 0;
    const int mode = (int)d[2] & ~VNET_PACK_ROUND_BF16, T = (int)d[3], I = (int)d[4], O = (int)d[5];
    const int CQ = (int)d[6], NP = (int)d[7];
    if (mode == VNET_PACK_FWD_X3 || mode == VNET_PACK_BWD_X3) {     // f32x3 images (conv_x3.h): CQ = k chunks, NP = n blocks of 16
        const uint32_t units = (uint32_t)CQ * X3_NPAIR * NP * 64;
        for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += gridDim.x * blockDim.x)
            x3_pack_unit(mode == VNET_PACK_BWD_X3, w, reinterpret_cast<u32x4*>(wp), I, O, NP, u);
        return;
    }
    if (mode == VNET_PACK_BOTH_X3) {
        // Both f32x3 images of a 5^3 filter from ONE read (round 6).  The per-unit form above reads w twice, the backward image in
        // 32-byte pieces (8 consecutive co of one ci per lane, lanes 1 KB apart): 879 MB of traffic at 3.5 TB/s for the network's 44 M
        // parameters.  Here, like the bf16 twin below: a workgroup stages one tap's [32 ci][32 co] fp32 slice in LDS (rows of 128
        // contiguous bytes) and emits its 128 forward units (8 consecutive ci of one co) and its 128 backward-data units (8 consecutive
        // co of one ci, at the flipped tap) -- the same units, the same exact split, the same image positions (bit-identical images:
        // tests/test_hip_parity_holes.py).  The empty half of the last pair (62) is zero-filled by the thread that holds the pair's only tap.
        __shared__ float sl[32][33];
        const uint32_t nci = (uint32_t)I / 32, nco = (uint32_t)O / 32, per_t = nci * nco, ntiles = 125u * per_t;
        const uint32_t ncobf = (uint32_t)O / 16, ncobb = (uint32_t)I / 16;      // n blocks of the forward / backward image
        u32x4* outf = reinterpret_cast<u32x4*>(wp);
        u32x4* outb = reinterpret_cast<u32x4*>(d[6]);
        const int tid = threadIdx.x;
        // (the next tile's four loads are issued before this tile's 24 stores: a workgroup keeps 4 KB of reads in flight under them)
        float nx[4];
        auto tile_load = [&](uint32_t tix) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, bi = rem / nco, bo = rem - bi * nco;
            const float* wt = w + ((size_t)t * I + bi * 32) * O + bo * 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) nx[j] = wt[(size_t)((tid >> 5) + 8 * j) * O + (tid & 31)];
        };
        if (blockIdx.x < ntiles) tile_load(blockIdx.x);
        for (uint32_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, bi = rem / nco, bo = rem - bi * nco;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[(tid >> 5) + 8 * j][tid & 31] = nx[j];
            __syncthreads();
            if (tix + gridDim.x < ntiles) tile_load(tix + gridDim.x);
            const uint32_t m = tid & 31, kg = (tid >> 5) & 3;           // unit: 8 k-values kg*8 .. of column / row m
            const bool fwd = tid < 128;
            const uint32_t tt = fwd ? t : 124u - t;                     // backward-data: the flipped tap
            int p, hi;
            x3_tap_pair((int)(tt / 25), (int)((tt / 5) % 5), (int)(tt % 5), p, hi);
            float v[8];
            if (fwd) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sl[kg * 8 + e][m];   // k = ci = bi*32 + kg*8 .., n = co = bo*32 + m
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sl[m][kg * 8 + e];   // k = co = bo*32 + kg*8 .., n = ci = bi*32 + m
            }
            const uint32_t kb = fwd ? bi : bo, nb = fwd ? bo : bi, ncob = fwd ? ncobf : ncobb;
            const uint32_t chunk = kb * 2 + (kg >> 1), half = kg & 1, cob = nb * 2 + (m >> 4), nl = m & 15;
            u32x4* out = fwd ? outf : outb;
            const size_t q = ((size_t)chunk * X3_NPAIR + p) * ncob + cob;
            x3_store_unit(out, q, nl + 16 * (half + 2 * hi), v);
            if (p == 62) {                                              // the pair's second half does not exist: zeros
                u32x4* dz0 = out + q * 3 * 64 + (nl + 16 * (half + 2));
                dz0[0] = dz0[64] = dz0[128] = u32x4{0u, 0u, 0u, 0u};
            }
            __syncthreads();
        }
        return;
    }
    if (mode == VNET_PACK_BOTH_BF16) {
        // Both bf16 images of a filter from ONE read (round 4): the two separate passes read every fp32 weight twice (352 MB for the
        // C5 network's 44 M parameters, + 176 MB of images).  A workgroup stages one tap's [32 ci][32 co] fp32 slice in LDS (rows of
        // 128 contiguous bytes) and emits its 128 forward units (8 consecutive ci of one co: a column walk) and its 128
        // backward-data units (8 consecutive co of one ci, at the flipped tap: a row walk) -- the same 16-byte units, the same
        // RNE rounding, the same image positions as the two modes below (bit-identical images: tests/test_hip_b16.py).
        __shared__ float sl[32][33];
        const uint32_t nci = (uint32_t)I / 32, nco = (uint32_t)O / 32, per_t = nci * nco, ntiles = (uint32_t)T * per_t;
        const uint32_t NPf = nco, NPb = nci;                 // cout blocks of the forward image, "cout" (= ci) blocks of the backward one
        u32x4* outf = reinterpret_cast<u32x4*>(wp);
        u32x4* outb = reinterpret_cast<u32x4*>(d[6]);
        const int tid = threadIdx.x;
        // (round 6: the next tile's four loads are issued before this tile's stores, as in the f32x3 twin above: 184 -> 126 us there)
        float nx[4];
        auto tile_load = [&](uint32_t tix) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, bi = rem / nco, bo = rem - bi * nco;
            const float* wt = w + ((size_t)t * I + bi * 32) * O + bo * 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) nx[j] = wt[(size_t)((tid >> 5) + 8 * j) * O + (tid & 31)];
        };
        if (blockIdx.x < ntiles) tile_load(blockIdx.x);
        for (uint32_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, bi = rem / nco, bo = rem - bi * nco;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[(tid >> 5) + 8 * j][tid & 31] = nx[j];
            __syncthreads();
            if (tix + gridDim.x < ntiles) tile_load(tix + gridDim.x);
            const uint32_t m = tid & 31, kg = (tid >> 5) & 3;           // unit: 8 k-values kg*8 .. of column / row m
            if (tid < 128) {
                // forward image [cin chunk][tap][cout block][cin half][32 cout][8 cin]: n = co = bo*32 + m, k = ci = bi*32 + kg*8 ..
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sl[kg * 8 + e][m];
                const uint32_t chunk = bi * 2 + (kg >> 1), hf = kg & 1;
                const u32x4 r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                outf[(((size_t)chunk * T + t) * NPf + bo) * 64 + hf * 32 + m] = r;
            } else {
                // backward image: tap T-1-t, n = ci = bi*32 + m, k = co = bo*32 + kg*8 ..
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sl[m][kg * 8 + e];
                const uint32_t chunk = bo * 2 + (kg >> 1), hf = kg & 1, tb = (uint32_t)T - 1 - t;
                const u32x4 r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                outb[(((size_t)chunk * T + tb) * NPb + bi) * 64 + hf * 32 + m] = r;
            }
            __syncthreads();
        }
        return;
    }
    if (mode == VNET_PACK_BOTH) {
        // the fp32 twin: forward image wp[t][ci/4][co][ci%4] and backward-data image wp[T-1-t][co/4][ci][co%4] from one read
        __shared__ float sl[32][33];
        const uint32_t nci = (uint32_t)I / 32, nco = (uint32_t)O / 32, per_t = nci * nco, ntiles = (uint32_t)T * per_t;
        float4* outf = reinterpret_cast<float4*>(wp);
        float4* outb = reinterpret_cast<float4*>(d[6]);
        const int tid = threadIdx.x;
        // (round 6: the next tile's four loads are issued before this tile's stores, as in the f32x3 twin above: 184 -> 126 us there)
        float nx[4];
        auto tile_load = [&](uint32_t tix) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, bi = rem / nco, bo = rem - bi * nco;
            const float* wt = w + ((size_t)t * I + bi * 32) * O + bo * 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) nx[j] = wt[(size_t)((tid >> 5) + 8 * j) * O + (tid & 31)];
        };
        if (blockIdx.x < ntiles) tile_load(blockIdx.x);
        for (uint32_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, bi = rem / nco, bo = rem - bi * nco;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[(tid >> 5) + 8 * j][tid & 31] = nx[j];
            __syncthreads();
            if (tix + gridDim.x < ntiles) tile_load(tix + gridDim.x);
            const uint32_t m = tid & 31, kg = tid >> 5;                 // 8 groups of 4 k-values x 32 columns / rows
            {
                const float4 v = make_float4(sl[kg * 4][m], sl[kg * 4 + 1][m], sl[kg * 4 + 2][m], sl[kg * 4 + 3][m]);
                outf[((size_t)t * (I / 4) + bi * 8 + kg) * O + bo * 32 + m] = v;
            }
            {
                const float4 v = make_float4(sl[m][kg * 4], sl[m][kg * 4 + 1], sl[m][kg * 4 + 2], sl[m][kg * 4 + 3]);
                outb[((size_t)(T - 1 - t) * (O / 4) + bo * 8 + kg) * I + bi * 32 + m] = v;
            }
            __syncthreads();
        }
        return;
    }
    if (mode == VNET_PACK_FWD_BF16 || mode == VNET_PACK_BWD_BF16) {      // here CQ = cin chunks, NP = cout blocks
        // one 16-byte unit (8 consecutive k of one n) per thread: consecutive lanes = consecutive n, so the forward image reads
        // 8 coalesced rows of the [I][O] slice and the backward image 32 contiguous bytes per lane; 32-bit index arithmetic
        // (a scalar 2-byte-per-thread version with 64-bit divisions ran at 2.7 TB/s: 196 us for the C5 network's filters)
        const uint32_t units = (uint32_t)CQ * T * NP * 64;
        u32x4* out = reinterpret_cast<u32x4*>(wp);
        for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += gridDim.x * blockDim.x) {
            const uint32_t m = u & 31, hf = (u >> 5) & 1;
            uint32_t q = u >> 6;
            const uint32_t cob = q % (uint32_t)NP; q /= (uint32_t)NP;
            const uint32_t t = q % (uint32_t)T, chunk = q / (uint32_t)T;
            const int k0 = (int)(chunk * 16 + hf * 8), n = (int)(cob * 32 + m);
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (mode == VNET_PACK_FWD_BF16) {
                if (n < O) {
                    const float* src = w + ((size_t)t * I + k0) * O + n;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (k0 + e < I) v[e] = src[(size_t)e * O];
                }
            } else {
                if (n < I) {
                    const float* src = w + ((size_t)(T - 1 - (int)t) * I + n) * O + k0;
                    if (k0 + 7 < O && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                        const float4 f0 = *reinterpret_cast<const float4*>(src), f1 = *reinterpret_cast<const float4*>(src + 4);
                        v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (k0 + e < O) v[e] = src[e];
                    }
                }
            }
            const u32x4 r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
            out[u] = r;
        }
        return;
    }
    if (mode == VNET_PACK_BWD) {
        // backward-data image = per tap the TRANSPOSE of the [I][O] filter slice in float4 column groups: staged through LDS
        // so that both the read (along O) and the write (along I) are >= 256 B contiguous per 16 lanes
        __shared__ float4 tile[16][65];
        const uint32_t tn = (NP + 63) / 64, tk = (CQ + 15) / 16, per_t = tn * tk, ntiles = (uint32_t)T * per_t;
        const int tid = threadIdx.x;
        float4* wp4 = reinterpret_cast<float4*>(wp);
        for (uint32_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
            const uint32_t t = tix / per_t, rem = tix - t * per_t, in_ = rem / tk, ik = rem - in_ * tk;
            const int n0 = (int)in_ * 64, c0 = (int)ik * 16;
            const float* wt = w + (size_t)(T - 1 - (int)t) * I * O;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (tid >> 4) + 16 * j, c4 = tid & 15, n = n0 + row, k0 = (c0 + c4) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < I && k0 < O) {
                    const float* src = wt + (size_t)n * O + k0;
                    if (k0 + 3 < O && (reinterpret_cast<uintptr_t>(src) & 15) == 0) v = *reinterpret_cast<const float4*>(src);
                    else { v.x = src[0]; if (k0 + 1 < O) v.y = src[1]; if (k0 + 2 < O) v.z = src[2]; if (k0 + 3 < O) v.w = src[3]; }
                }
                if (r16) { const uint32_t a0 = pk_bf16(v.x, v.y), a1 = pk_bf16(v.z, v.w); v = make_float4(bf_lo(a0), bf_hi(a0), bf_lo(a1), bf_hi(a1)); }
                tile[c4][row] = v;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = (tid >> 6) + 4 * j, row = tid & 63, n = n0 + row, cq = c0 + c4;
                if (n < NP && cq < CQ) wp4[((size_t)t * CQ + cq) * NP + n] = tile[c4][row];
            }
            __syncthreads();
        }
        return;
    }
    // one float4 of the packed image (4 consecutive k of one n) per thread, 32-bit index arithmetic: the per-element
    // 64-bit divisions of a scalar version made this launch ALU-bound (245 us for 44 M parameters)
    const uint32_t total4 = (uint32_t)(mode == VNET_PACK_UP ? 1 : T) * CQ * NP;
    const uint32_t uNP = NP, uCQ = CQ;
    for (uint32_t o4 = blockIdx.x * blockDim.x + threadIdx.x; o4 < total4; o4 += gridDim.x * blockDim.x) {
        const uint32_t q = o4 / uNP, n = o4 - q * uNP;
        const uint32_t t = q / uCQ, cq = q - t * uCQ;
        const int k0 = (int)cq * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (mode == VNET_PACK_FWD) {
            if ((int)n < O) {
                const float* src = w + ((size_t)t * I + k0) * O + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (k0 + r < I) v[r] = src[(size_t)r * O];
            }
        } else if (mode == VNET_PACK_BWD) {
            if ((int)n < I) {
                const float* src = w + ((size_t)(T - 1 - (int)t) * I + n) * O + k0;
                if ((O & 3) == 0 && k0 + 3 < O && (reinterpret_cast<uintptr_t>(src) & 15) == 0) { const float4 f = *reinterpret_cast<const float4*>(src); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (k0 + r < O) v[r] = src[r];
                }
            }
        } else {
            if ((int)n < 8 * O) {
                const float* src = w + (size_t)n * I + k0;
                if ((I & 3) == 0 && k0 + 3 < I && (reinterpret_cast<uintptr_t>(src) & 15) == 0) { const float4 f = *reinterpret_cast<const float4*>(src); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (k0 + r < I) v[r] = src[r];
                }
            }
        }
        if (r16) { const uint32_t a0 = pk_bf16(v[0], v[1]), a1 = pk_bf16(v[2], v[3]); v[0] = bf_lo(a0); v[1] = bf_hi(a0); v[2] = bf_lo(a1); v[3] = bf_hi(a1); }
        reinterpret_cast<float4*>(wp)[o4] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

void packed_dims(int mode, int T, int I, int O, int* Tp, int* CQ, int* NP) {
    if (mode == VNET_PACK_FWD) { *Tp = T; *CQ = round_up(I, 16) / 4; *NP = round_up(O, 16); }
    else if (mode == VNET_PACK_BWD) { *Tp = T; *CQ = round_up(O, 16) / 4; *NP = round_up(I, 16); }
    else { *Tp = 1; *CQ = round_up(I, 16) / 4; *NP = round_up(8 * O, 16); }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
// The dynamic-LDS attribute is per (kernel, device): remember per device which ones are configured (a process that
// drives a non-zero device, or several, must configure each of them once).
template <typename K>
int ensure_lds(K kernel, size_t bytes, unsigned long long& done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    // (atomic: launches may come from several host threads -- vnet_infer's workers, torch's autograd thread; setting the
    // attribute twice is harmless, losing another device's bit to a torn read-modify-write would only repeat it)
    if (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit) return 0;
    const int e = set_lds(kernel, bytes);
    if (e == 0) __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
    return e;
}

// compute units of the current device (cached per device; 256 on MI355X)
inline int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cached[dev & 63];
    if (c == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        c = n;
    }
    return c;
}

int pick_ns(int CoutP) { return (CoutP % 64 == 0) ? 4 : (CoutP % 32 == 0) ? 2 : 1; }

struct ConvPlan { int ns, ncob, nbz, nby, nbx, nsplit, cps, small, nz, half, tiny; };

template <int TZ, int TY, int TX>
void brick_counts(int Do, int Ho, int Wo, ConvPlan& p) { p.nbz = ceil_div(Do, TZ); p.nby = ceil_div(Ho, TY); p.nbx = ceil_div(Wo, TX); }

// brick shapes: 5^3 stride-1 convs on W >= 16 use 4x8x8 bricks with 4-wave workgroups: the 74 KB tile lets TWO workgroups
// share a CU, so one's tile staging and barriers overlap the other's MFMAs (+2..8 % over a 4x8x16 brick with 8 waves
// and one workgroup per CU, measured); W < 16 uses the 8x8x8 cube, the 2^3 kernels their own shapes
ConvPlan plan_conv(int ks, int stride, int up, int Cin, int Cout, int B, int Do, int Ho, int Wo, int gridW) {
    ConvPlan p{};
    const int CoutP = up ? round_up(8 * Cout, 16) : round_up(Cout, 16);
    p.ns = pick_ns(CoutP);
    p.ncob = CoutP / (16 * p.ns);
    p.small = gridW < 16;
    // 2^3 stride-2 conv, W >= 16: 1x4x16 output bricks (32 KB input tile, four workgroups per CU) instead of 2x4x16: these launches are
    // load latency / bandwidth, more tiles in flight per CU help (128^3 -> 64^3: 47.8 -> 38.8 us, profiles/bench_updown.py)
    if (stride == 2 && !up) { if (p.small) brick_counts<2, 8, 8>(Do, Ho, Wo, p); else brick_counts<1, 4, 16>(Do, Ho, Wo, p); }
    else if (ks == 5 && !up && !p.small) { p.half = 1; brick_counts<4, 8, 8>(Do, Ho, Wo, p); }
    else if (ks == 5 && !up && tuning().f32_small >= 2 && Do <= 4 && Ho <= 4 && Wo <= 4 && Cin >= 32) { p.half = 2; brick_counts<4, 4, 4>(Do, Ho, Wo, p); }      // (Cin >= 32: never the 5x5x1 input block)
    else if (ks == 5 && !up && tuning().f32_small >= 1) { p.half = 1; brick_counts<4, 8, 8>(Do, Ho, Wo, p); }
    else if (up && !p.small) brick_counts<2, 4, 16>(Do, Ho, Wo, p);       // transposed conv: 128 input voxels per 4-wave workgroup (55.5 -> 52.5 us at 128^3)
    else { if (p.small) brick_counts<8, 8, 8>(Do, Ho, Wo, p); else brick_counts<4, 8, 16>(Do, Ho, Wo, p); }
    const int nchunks = round_up(Cin, 16) / 16;
    // deep levels have few bricks: prefer more, narrower cout blocks (each a full workgroup of equal work) until
    // bricks x cout-blocks x channel-chunks fills the 256 CUs in one round, before resorting to tap splits
    if (ks == 5 && !up) {
        const long nb = (long)B * p.nbz * p.nby * p.nbx;
        const int ncob1 = p.ns * p.ncob;                      // cout blocks at NS = 1
        // (4x4x4 bricks: a workgroup is four waves with one 16-voxel subtile each and a 32 KB tile -- several share a CU, so aim for two
        //  rounds of workgroups: 4^3 x 2 256->256 34.3 -> 29.9 us; on the 4x8x8 bricks of the 8^3 volumes it cost 2 %)
        const long fill = p.half == 2 ? 512 : 256;
        if (nb * p.ncob < fill && nb * ncob1 >= fill) {       // fill the chip WITHOUT split-K if a narrower block can
            while (nb * p.ncob < fill) { p.ns /= 2; p.ncob *= 2; }
        } else {
            while (p.ns > 1 && nb * p.ncob * nchunks < fill) { p.ns /= 2; p.ncob *= 2; }
        }
    }
    if (ks == 2 && !up) {                                     // 2^3 stride-2 conv at the coarse levels: narrower cout blocks first
        const long nb = (long)B * p.nbz * p.nby * p.nbx;
        while (p.ns > 1 && nb * p.ncob < 256) { p.ns /= 2; p.ncob *= 2; }
    }
    if (up) {
        // transposed 2^3 conv = one GEMM with N = 8*Cout: the coarse levels have one or two bricks, so take narrower
        // column blocks and, for W < 16, 2x8x8 bricks (4 waves) until the launch has ~256 workgroups
        long nb = (long)B * p.nbz * p.nby * p.nbx;
        while (p.ns > 1 && nb * p.ncob < 256) { p.ns /= 2; p.ncob *= 2; }
        if (p.small && nb * p.ncob < 256) { p.tiny = 1; brick_counts<2, 8, 8>(Do, Ho, Wo, p); }
    }
    const int nwg = B * p.nbz * p.nby * p.nbx * p.ncob;
    p.nsplit = 1;
#ifdef VNET_PLAN_ENV
    static const int tgt = getenv("VNET_F32_SPLIT_TARGET") ? atoi(getenv("VNET_F32_SPLIT_TARGET")) : 512;
    static const int nzmin = getenv("VNET_F32_NZ_MIN") ? atoi(getenv("VNET_F32_NZ_MIN")) : 256;
#else
    constexpr int tgt = 512, nzmin = 256;
#endif
    if (!up && nwg < 256 && nchunks > 1) p.nsplit = min(nchunks, ceil_div(tgt, nwg));
    p.cps = ceil_div(nchunks, p.nsplit);
    p.nsplit = ceil_div(nchunks, p.cps);
    p.nz = (ks == 5 && !up && nwg * p.nsplit < nzmin) ? 5 : 1;
    return p;
}

template <int KS, int STRIDE, int TZ, int TY, int TX, int WAVES, int MS, bool UP, int KX = KS, bool STATS = false, bool IO16 = false>
int launch_conv_ns(const ConvArgs& a, const ConvPlan& p, hipStream_t st) {
    using G = TileGeom<KS, STRIDE, TZ, TY, TX, KX>;
    const size_t lds = (size_t)G::LDS_FLOATS * 4;
    dim3 grid(a.B * p.nbz * p.nby * p.nbx, p.ncob, p.nsplit * p.nz), block(WAVES * 64);
    int e = 0;
#define VNET_GO(NSV)                                                                              \
    {                                                                                             \
        auto k = conv_kernel<KS, STRIDE, TZ, TY, TX, WAVES, MS, NSV, UP, KX, STATS, IO16>;            \
        static unsigned long long attr_done = 0;                                                  \
        if (int ae = ensure_lds(k, lds, attr_done)) return ae;                                    \
        hipLaunchKernelGGL(k, grid, block, lds, st, a);                                           \
    }
    if (p.ns == 4) VNET_GO(4) else if (p.ns == 2) VNET_GO(2) else VNET_GO(1)
#undef VNET_GO
    e = (int)hipGetLastError();
    return e;
}


// ------------------------------------------------------------------------------------------
// bf16-operand variant of the 5x5x5 convolution (BASELINE config C5: "bf16 compute, fp32 accumulate").
// Activations stay fp32 NDHWC in HBM; they are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while the brick
// is staged into LDS, the filter is packed as bf16, and v_mfma_f32_32x32x16_bf16 accumulates in fp32:
//   D[row = cout (32)][col = voxel (32)] += A[cout][k = 16 cin] * B[k][voxel]
//   * same 512-voxel brick as the fp32 kernel; 8 waves x MS=2 subtiles of 32 voxels, NSB blocks of 32 cout;
//   * LDS tile = two planes [cin half][voxel][8 bf16]: a B fragment is ONE 16-byte ds_read_b128 whose
//     address is lane_base + compile-time tap offset; voxel order inside a subtile is rotated on the second
//     x row so that every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) covers 16 distinct 16-byte slots;
//   * at bf16 rates the filter can no longer stream from L2 per wave (64 B/clk/CU): one dz plane of it
//     (25 taps x NSB KB) is staged in LDS, the next plane prefetched into registers during the MFMAs.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TZ, int TY, int TX>
struct Bf16Geom {
    static constexpr int IZ = TZ + 4, IY = TY + 4, IX = TX + 4;
    static constexpr int NV = IZ * IY * IX;
    static constexpr int PLANE = NV * 16;                  // bytes of one cin-half plane
    static constexpr int TILE_BYTES = 2 * PLANE;
};

// 64 bytes of zeros in device memory: masked-off lanes of a register prefetch load from HERE (an address select) instead of
// loading anywhere and selecting the DATA afterwards -- a data select makes the wave wait for the load right where it was
// issued (hipcc emitted s_waitcnt vmcnt(0) in front of the MFMA loop: measured 4-5 K cycles per brick step, s_memtime stamps),
// i.e. the "prefetch" was synchronous.
__device__ __attribute__((aligned(64))) const unsigned int vnet_zero_line[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
// (explicit global address space: with a plain pointer select hipcc falls back to flat_load, which also counts on lgkmcnt
// and would then be waited for by the first LDS fragment read of the MFMA loop)
typedef const __attribute__((address_space(1))) u32x4* gvec16_t;
__device__ __forceinline__ u32x4 load16_or_zero(const unsigned short* p, bool ok) {
    gvec16_t src = ok ? (gvec16_t)(p) : (gvec16_t)(vnet_zero_line);
    return *src;
}

// bf16-source twin of XTile (round 2: activations that carry a bf16 shadow, see vnet_conv_fwd_bf16_x16): a thread owns one
// (x, cin half) column = 8 channels = ONE 16-byte unit per row, loaded as it will sit in LDS -- half the bytes through L2,
// half the load/store instructions and no conversion.  Channel counts must be multiples of 8.
template <int IZ, int IY, int IX, int NT>
struct XTileH {
    static constexpr int COLS = IX * 2;
    static constexpr int RPI = NT / COLS;
    static constexpr int ROWS = IZ * IY;
    static constexpr int PER = (ROWS + RPI - 1) / RPI;
    static_assert(RPI >= 1, "tile row wider than the workgroup");
    template <int K0, int KN>
    __device__ static __forceinline__ void issue_part(u32x4 (&v)[KN], const unsigned short* __restrict__ x0, const unsigned short* __restrict__ x1,
                                                      int C0, int C1, int chunk, int b, int gz0, int gy0, int gx0,
                                                      int Di, int Hi, int Wi, int tid) {
        int r0 = tid / COLS;
        const int col = tid - r0 * COLS;
        // r0 is made opaque per call: everything below that depends only on the thread (the per-row offsets: 2 x PER registers as
        // 64-bit values) would otherwise be hoisted out of the caller's brick loop and held for the whole kernel -- in the
        // row-pair kernel that spilled, and the scratch reloads between the loads made the prefetch synchronous (round-3 stamps:
        // 9-10 K cycles of tile issue per step instead of 2-3 K)
        asm volatile("" : "+v"(r0));
        const int ix = col >> 1, hf = col & 1;
        const int c = chunk * 16 + hf * 8;
        const int gx = gx0 + ix;
        const bool colok = r0 < RPI && (unsigned)gx < (unsigned)Wi && c < C0 + C1;
        const bool first = c < C0 || !colok;
        const int Cs = first ? C0 : C1;
        const int rowstride = Wi * Cs, planestride = Hi * rowstride;
        const unsigned short* bp = (first ? x0 + (colok ? c : 0) : x1 + (c - C0)) + (size_t)b * Di * planestride + (colok ? gx * Cs : 0);
        constexpr int DIZ = RPI / IY, DIY = RPI % IY;
        int row = r0 + K0 * RPI;
        int iz = row / IY, iy = row - iz * IY;
        // Interior bricks (halo and channel chunk entirely inside the tensors -- 62 % of the bricks of a 128^3 volume): nothing
        // to mask, so a load is an offset add.  The address arithmetic of the general path costs ~27 instructions per load
        // (s_memtime stamps: 2 K cycles per brick step for 8 loads in the 16-cout kernel) at a moment when the other wave of the
        // SIMD is doing exactly the same, i.e. with the matrix pipe idle.  (Wave-uniform branch: every lane sees the same brick.)
        const int csu = (chunk * 16 < C0) ? C0 : C1;                                      // (uniform: a chunk never straddles the sources)
        const bool interior = gz0 >= 0 && gz0 + IZ <= Di && gy0 >= 0 && gy0 + IY <= Hi && gx0 >= 0 && gx0 + IX <= Wi &&
                              (chunk + 1) * 16 <= C0 + C1 && (C0 & 15) == 0 && (long long)Di * Hi * Wi * csu < (1ll << 31);
        if (interior) {
            // one UNIFORM 64-bit base (source tensor, sample) + a 32-bit element offset per load (saddr form of global_load)
            const unsigned short* src = ((chunk * 16 < C0) ? x0 + chunk * 16 : x1 + (chunk * 16 - C0)) + (size_t)b * Di * Hi * Wi * csu;
            const int rs = Wi * csu;
            const int r0c = min(r0, RPI - 1);                                              // idle threads (r0 >= RPI) load a valid row, commit drops it
            const int col0 = (gx0 + ix) * csu + hf * 8;
            int rowc = r0c + K0 * RPI;
            int jz = rowc / IY, jy = rowc - jz * IY;
#pragma unroll
            for (int k = 0; k < KN; ++k) {
                const unsigned off = (unsigned)(((gz0 + min(jz, IZ - 1)) * Hi + gy0 + jy) * rs + col0);   // rows past the tile (last iteration) stay inside it
                v[k] = *(gvec16_t)(src + off);
                jy += DIY; jz += DIZ;
                if (jy >= IY) { jy -= IY; ++jz; }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            const int gz = gz0 + iz, gy = gy0 + iy;
            const bool ok = colok && row < ROWS && (unsigned)gz < (unsigned)Di && (unsigned)gy < (unsigned)Hi;
            v[k] = load16_or_zero(bp + (ok ? gz * planestride + gy * rowstride : 0), ok);
            row += RPI; iy += DIY; iz += DIZ;
            if (iy >= IY) { iy -= IY; ++iz; }
        }
    }
};

// filter-plane prefetch: WPER 16-byte units per thread, global -> registers (issue) -> LDS (commit)
template <int NSB, int WPER, int NT>
__device__ __forceinline__ void bf16_w_issue(u32x4 (&wreg)[WPER], const u32x4* __restrict__ src, int ncob, int tid) {
    constexpr int WUNITS = 25 * NSB * 64;
#pragma unroll
    for (int k = 0; k < WPER; ++k) {
        const int idx = min(tid + k * NT, WUNITS - 1);
        const int t = idx / (NSB * 64), j = idx - t * (NSB * 64);
        wreg[k] = src[(size_t)t * ncob * 64 + j];
    }
}
template <int NSB, int WPER, int NT>
__device__ __forceinline__ void bf16_w_commit(u32x4* wl, const u32x4 (&wreg)[WPER], int tid) {
    constexpr int WUNITS = 25 * NSB * 64;
#pragma unroll
    for (int k = 0; k < WPER; ++k) {
        const int idx = tid + k * NT;
        wl[idx < WUNITS ? idx : WUNITS] = wreg[k];          // WUNITS = one spare 16-byte dump slot behind the slab
    }
}
// brick (+halo) of one 16-channel chunk, bf16 source: 16-byte units as they sit in LDS, two planes [cin half][voxel][8]
template <typename G, typename XH, int K0, int KN>
__device__ __forceinline__ void bf16_tile_commit_h(unsigned char* tile, unsigned char* dump, const u32x4 (&v)[KN], int tid) {
    const int r0 = tid / XH::COLS, col = tid - r0 * XH::COLS;
    unsigned char* base = tile + (col & 1) * G::PLANE + (col >> 1) * 16;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        const int row = r0 + (K0 + k) * XH::RPI;
        const bool ok = r0 < XH::RPI && row < XH::ROWS;
        *reinterpret_cast<u32x4*>(ok ? base + row * (G::IX * 16) : dump) = v[k];
    }
}

// bf16-storage mode (round 3; the only form since round 5 retired the fp32-source / fp32-output instantiations, whose template paths were
// deleted in round 6): x0 / x1 are bf16 tensors (same NDHWC indexing, 2-byte elements), and so are the outputs, the accumulate source and
// the residual (epilogue_b16_batch)
template <int TZ, int TY, int TX, int NSB, int WAVES, bool STATS = false>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_bf16_kernel(ConvArgs a) {
    using G = Bf16Geom<TZ, TY, TX>;
    constexpr int MS = 2, NT = WAVES * 64;
    static_assert(TZ * TY * TX == WAVES * MS * 32, "brick = WAVES*MS subtiles of 32 voxels");
    constexpr int WUNITS = 25 * NSB * 64;                  // 16-byte units of one dz plane of the filter slab
    constexpr int WPER = (WUNITS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    u32x4* wl = reinterpret_cast<u32x4*>(smem + G::TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p32 = lane & 31, half = lane >> 5;
    // voxel of this lane inside a 32-voxel subtile (see header: bank-conflict-free rotation of the 2nd row)
    int q32;
    if (TX == 16) {
        q32 = (p32 >= 16) ? 16 + ((p32 + 12) & 15) : p32;
    } else {
        // TX == 8 (row pitch 12 voxels): a subtile is 4 rows of 8; ds_read_b128 lane group {0-3,12-15,20-27} takes rows
        // 0 and 2, group {4-11,16-19,28-31} rows 1 and 3 -> tile indices 0..7,24..31 / 12..19,36..43 are distinct mod 16
        const bool ga = p32 < 4 || (p32 >= 12 && p32 < 16) || (p32 >= 20 && p32 < 28);
        const int j = ga ? (p32 < 4 ? p32 : p32 < 16 ? p32 - 8 : p32 - 12) : (p32 < 12 ? p32 - 4 : p32 < 20 ? p32 - 8 : p32 - 16);
        q32 = ((j >> 3) * 2 + (ga ? 0 : 1)) * 8 + (j & 7);
    }

    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    int brick = xcd_remap(blockIdx.x, nbrick);
    const int brick_id = brick;
    const int bx = brick % a.nbx; brick /= a.nbx;
    const int by = brick % a.nby; brick /= a.nby;
    const int bz = brick % a.nbz; const int b = brick / a.nbz;
    const int ncob = a.CoutP / 32;
    const int cob0 = blockIdx.y * NSB;
    const int co0 = cob0 * 32;
    const int split = blockIdx.z / a.nz, zsplit = blockIdx.z - split * a.nz;
    const int dz0 = zsplit * 5 / a.nz, dz1 = (zsplit + 1) * 5 / a.nz;
    const int c_begin = split * a.cps;
    const int c_end = min(a.nchunks, c_begin + a.cps);

    int boff[MS];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int v = (wave * MS + m) * 32 + q32;
        const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
        boff[m] = half * G::PLANE + ((vz * G::IY + vy) * G::IX + vx) * 16;
    }
    const int aoff = half * 32 + p32;                       // uint4 units inside one (tap, cout block) KB

    f32x16 acc[MS][NSB];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int n = 0; n < NSB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int gz0 = bz * TZ - 2, gy0 = by * TY - 2, gx0 = bx * TX - 2;
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp);

    u32x4 wreg[WPER];
    unsigned char* dump = smem + G::TILE_BYTES + WUNITS * 16 + (tid & 63) * 16;       // per-lane slot for masked-off stores
    auto wsrc = [&](int chunk, int dz) { return wg + ((size_t)(chunk * 125 + dz * 25) * ncob + cob0) * 64; };
    auto stage_tile = [&](int chunk) {
        using XH = XTileH<G::IZ, G::IY, G::IX, NT>;
        u32x4 v[XH::PER];
        XH::template issue_part<0, XH::PER>(v, reinterpret_cast<const unsigned short*>(a.x0), reinterpret_cast<const unsigned short*>(a.x1),
                                            a.C0, a.C1, chunk, b, gz0, gy0, gx0, a.Di, a.Hi, a.Wi, tid);
        __builtin_amdgcn_sched_barrier(0);
        bf16_tile_commit_h<G, XH, 0, XH::PER>(tile, dump, v, tid);
    };

    if (c_begin < c_end) {
        bf16_w_issue<NSB, WPER, NT>(wreg, wsrc(c_begin, dz0), ncob, tid);
        stage_tile(c_begin);
        bf16_w_commit<NSB, WPER, NT>(wl, wreg, tid);
    }
    __syncthreads();

    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        for (int dz = dz0; dz < dz1; ++dz) {
            // prefetch the next filter plane (this chunk's next dz, or the next chunk's first) into registers
            const bool last_dz = (dz + 1 == dz1);
            const bool more = !(last_dz && chunk + 1 == c_end);
            if (more) {
                bf16_w_issue<NSB, WPER, NT>(wreg, wsrc(last_dz ? chunk + 1 : chunk, last_dz ? dz0 : dz + 1), ncob, tid);
                __builtin_amdgcn_sched_barrier(0);
            }

            const unsigned char* tp = tile + dz * (G::IY * G::IX * 16);
            const u32x4* wa = wl + aoff;
            bf16x8 bf[2][MS], af[2][NSB];
#pragma unroll
            for (int m = 0; m < MS; ++m) bf[0][m] = *reinterpret_cast<const bf16x8*>(tp + boff[m]);
#pragma unroll
            for (int n = 0; n < NSB; ++n) af[0][n] = *reinterpret_cast<const bf16x8*>(wa + n * 64);
            // taps of a plane in (dx, dy) order -- the order of the row-pair kernel below, so that both kernels add the
            // same products in the same sequence (bit-identical results whichever one a launch takes)
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                if (t + 1 < 25) {
                    const int tn = t + 1;
                    const int tap = (tn % 5) * 5 + tn / 5;                  // dy = tn % 5, dx = tn / 5
                    const int o = ((tn % 5) * G::IX + (tn / 5)) * 16;
#pragma unroll
                    for (int m = 0; m < MS; ++m) bf[tn & 1][m] = *reinterpret_cast<const bf16x8*>(tp + boff[m] + o);
#pragma unroll
                    for (int n = 0; n < NSB; ++n) af[tn & 1][n] = *reinterpret_cast<const bf16x8*>(wa + (tap * NSB + n) * 64);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int n = 0; n < NSB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][n], bf[t & 1][m], acc[m][n], 0, 0, 0);
            }
            __syncthreads();                       // every wave is done with this filter plane (and, on the last dz, the tile)
            if (more) {
                if (last_dz) stage_tile(chunk + 1);
                bf16_w_commit<NSB, WPER, NT>(wl, wreg, tid);
            }
            __syncthreads();
        }
    }

    // epilogue: register r of lane = cout co0 + n*32 + 8*(r/4) + 4*half + r%4 of voxel (m, q32)
    constexpr int NSS = STATS ? NSB : 1;
    float s1[NSS][4][4], s2[NSS][4][4];          // batch-norm statistics of this lane's channels (STATS)
#pragma unroll
    for (int n = 0; n < NSS; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) s1[n][g][k] = s2[n][g][k] = 0.f;
    {
        if (!a.part) {
            // bf16 outputs, no split-K: the four channel groups of a voxel and cout block as one batch (epilogue_b16_batch)
#pragma unroll
            for (int n = 0; n < NSB; ++n) {
                float bq[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co0 + n * 32 + g * 8 + half * 4;
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.bias && co < a.Cout) bv = make_float4(a.bias[co], a.bias[co + 1], a.bias[co + 2], a.bias[co + 3]);
                    bq[g][0] = bv.x; bq[g][1] = bv.y; bq[g][2] = bv.z; bq[g][3] = bv.w;
                }
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    const int v = (wave * MS + m) * 32 + q32;
                    const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
                    const int oz = bz * TZ + vz, oy = by * TY + vy, ox = bx * TX + vx;
                    const bool vok = oz < a.Do && oy < a.Ho && ox < a.Wo;
                    const size_t ov = vok ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
                    size_t ovs[4]; int cos[4]; bool oks[4]; float e[4][4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = co0 + n * 32 + g * 8 + half * 4;
                        oks[g] = vok && co < a.Cout;
                        ovs[g] = oks[g] ? ov : 0;
                        cos[g] = oks[g] ? co : 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) e[g][k] = acc[m][n][g * 4 + k] + bq[g][k];
                    }
                    epilogue_b16_batch<STATS, 4>(a, ovs, cos, oks, e);
                    if constexpr (STATS) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (oks[g]) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) { s1[n][g][k] += e[g][k]; s2[n][g][k] += e[g][k] * e[g][k]; }
                            }
                    }
                }
            }
        }
    }
    if (a.part) {                              // split K: the raw fp32 partial sums; bias / accumulate / rounding / statistics belong to the reduce
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            const int v = (wave * MS + m) * 32 + q32;
            const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
            const int oz = bz * TZ + vz, oy = by * TY + vy, ox = bx * TX + vx;
            if (oz >= a.Do || oy >= a.Ho || ox >= a.Wo) continue;
            const size_t ov = ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
#pragma unroll
            for (int n = 0; n < NSB; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co0 + n * 32 + g * 8 + half * 4;
                    *reinterpret_cast<float4*>(a.part + blockIdx.z * a.part_stride + ov * a.CoutP + co) =
                        make_float4(acc[m][n][g * 4], acc[m][n][g * 4 + 1], acc[m][n][g * 4 + 2], acc[m][n][g * 4 + 3]);
                }
        }
    }
    if constexpr (STATS) if (!a.part) {
        // the 32 lanes of a half hold the same channels for 32 voxels: butterfly over p32, then across waves via LDS
        // (the main loop ends with a barrier after the last tile read, so the LDS is free here)
        constexpr int CW = NSB * 32;
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int n = 0; n < NSB; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[n][g][k] = half32_sum(s1[n][g][k]);
                    s2[n][g][k] = half32_sum(s2[n][g][k]);
                    if (p32 == 0) {
                        red[wave * 2 * CW + n * 32 + g * 8 + half * 4 + k] = s1[n][g][k];
                        red[wave * 2 * CW + CW + n * 32 + g * 8 + half * 4 + k] = s2[n][g][k];
                    }
                }
        __syncthreads();
        stats_row_write<WAVES, CW>(red, a.stats, (size_t)brick_id, co0, a.Cout, tid);
    }
}

// ------------------------------------------------------------------------------------------
// 16-output-channel variant (round 2).  The kernel above computes D[32 cout][32 voxels]; layers with 16 output channels
// (every 5^3 conv at full resolution: 4->16, 16->16, 32->16 forward, 16->16 backward-data) pad to 32 and waste half of
// their MFMAs.  Here:  v_mfma_f32_16x16x32_bf16,  D[16 cout][16 voxels] += A[16 cout][k = 2 taps x 16 cin] * B[k][16 voxels]
//   * a subtile = one x row of 16 voxels; a wave owns 4 rows adjacent in y, so the B fragment of (row m, tap dy) IS the
//     fragment of (row m+1, tap dy-1): per (dz pair, dx) a wave reads 8 row fragments for 20 MFMAs (2.5x fewer LDS bytes
//     than one read per MFMA, which would make this shape LDS-bound);
//   * K = 32 pairs two taps that differ by a constant LDS offset (dz, dz+1 -> one tile plane; for dz = 4: dy, dy+1 -> one
//     tile row), so lanes 32-63 just use a base address shifted by that constant; 260 MFMAs per subtile-quad instead of 250;
//   * the whole 16-cin chunk of the filter (65 fragments x 1 KB, compacted from the generic packed image: only the 16 real
//     cout) stays in LDS -- no per-plane barriers; for Cin = 16 it is loaded once per workgroup;
//   * persistent workgroups (one per CU) walk their bricks; the next tile (and filter chunk) is prefetched global ->
//     registers during the MFMAs and committed between two barriers.
// ------------------------------------------------------------------------------------------
// IN4 (round 3): the network input of a multi-modality net -- 4 real channels zero-padded to 8.  With K = 2 taps x 16 zero-padded
// channels three quarters of every MFMA multiply zeros; here the 16 K-channels of a tap pair are j = (sx, c): x-SHIFT sx = 0..3
// times modality c = 0..3 (an x-im2col done while the tile is committed to LDS: every loaded voxel's 8 bytes go to the four
// units (x - sx, slot sx)), so one MFMA covers the taps dx = 0..3 of a (dz pair, dy) and a second one, at x offset 4, the tap
// dx = 4 (its other twelve K-channels meet zero weights): 26 filter fragments and 104 MFMAs per 4 rows instead of 65 and 260.
// (bf16 tensors in and out: the fp32-source / fp32-output template paths of rounds 2-4 were deleted in round 6)
template <int TZ, int TY, int TX, bool STATS = false, bool IN4 = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_bf16_c16_kernel(ConvArgs a) {
    using G = Bf16Geom<TZ, TY, TX>;
    static_assert(TZ == 4 && TY == 8 && TX == 16, "8 waves x 4 rows of 16 voxels");
    constexpr int NDX = IN4 ? 2 : 5, XSTEP = IN4 ? 4 : 1;                 // x groups of a (dz pair, dy) and their tile offset
    constexpr int NT = 512, NFRAG = IN4 ? 26 : 65, FUNITS = NFRAG * 64, FPER = (FUNITS + NT - 1) / NT;
    constexpr int ROWB = G::IX * 16, PLANEB = G::IY * G::IX * 16;       // bytes per tile row / per tile z-plane (one cin half)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    unsigned char* fl = smem + G::TILE_BYTES;                            // [65 fragments][64 lanes][16 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4, half = g & 1, hi = g >> 1;
    const int vz = wave >> 1, vy0 = (wave & 1) * 4;
    unsigned char* dump = smem + G::TILE_BYTES + NFRAG * 1024 + lane * 16;
    float* red = reinterpret_cast<float*>(smem + G::TILE_BYTES + NFRAG * 1024 + 64 * 16);      // [8 waves][2 x 16] epilogue statistics

    const int base0 = half * G::PLANE + ((vz * G::IY + vy0) * G::IX + j) * 16;
    const unsigned char* bZ = tile + base0 + hi * PLANEB;                // taps (dz, dz+1)
    const unsigned char* bY = tile + base0 + hi * ROWB;                  // taps (4, dy), (4, dy+1)
    const unsigned char* b0 = tile + base0;                              // single tap (4, 4): both lane halves read the same row
    const unsigned char* fa = fl + lane * 16;

    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    const int G8 = gridDim.x >> 3;                                       // workgroups per XCD (grid is a multiple of 8)
    const int per_xcd = (nbrick + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b_lo = xcd * per_xcd, b_hi = min(nbrick, b_lo + per_xcd);
    const int nmine = (b_hi - b_lo - slot + G8 - 1) / G8;                // bricks b_lo + slot + i * G8
    if (b_lo + slot >= b_hi) return;
    const int nch = a.nchunks;
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp);

    auto brick_origin = [&](int i, int& b, int& bz, int& by, int& bx) {
        int brick = b_lo + slot + i * G8;
        bx = brick % a.nbx; brick /= a.nbx;
        by = brick % a.nby; brick /= a.nby;
        bz = brick % a.nbz; b = brick / a.nbz;
    };
    // source (16-byte unit of the generic packed image, ncob = 1) of filter fragment unit u = f * 64 + l
    auto fsrc = [&](int u, int chunk, bool& valid) -> const u32x4* {
        const int f = u >> 6, l = u & 63;
        const int co = l & 15, gg = l >> 4, hf = gg & 1, up = gg >> 1;
        int dz, dy, dx;
        valid = u < FUNITS;
        if (f < 50) { const int zp = f / 25, r = f - zp * 25; dx = r / 5; dy = r - dx * 5; dz = 2 * zp + up; }
        else { const int r = f - 50; dx = r / 3; const int q = r - dx * 3; dz = 4; dy = 2 * q + up; if (q == 2 && up) valid = false; }
        const int tap = valid ? (dz * 5 + dy) * 5 + dx : 0;
        return wg + ((size_t)(chunk * 125 + tap) * 2 + hf) * 32 + co;
    };
    u32x4 freg[FPER];
    auto filter_issue = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < FPER; ++k) {
            if constexpr (IN4) {
                // unit (fragment f, lane l): 8 K-channels j = 8 hf .. 8 hf + 7 = x shifts 2 hf, 2 hf + 1 x 4 modalities of cout co:
                // the first 8 bytes (cin 0..3) of the generic image's units of the taps dx = 4 dxg + 2 hf and dx + 1
                const int u = min(tid + k * NT, FUNITS - 1);
                const int f = u >> 6, l = u & 63;
                const int co = l & 15, gg = l >> 4, hf = gg & 1, up = gg >> 1;
                int dz, dy, dxg;
                bool valid = tid + k * NT < FUNITS;
                if (f < 20) { const int zp = f / 10, r = f - zp * 10; dxg = r / 5; dy = r - dxg * 5; dz = 2 * zp + up; }
                else { const int r = f - 20; dxg = r / 3; const int q = r - dxg * 3; dz = 4; dy = 2 * q + up; if (q == 2 && up) valid = false; }
                const int dxa = 4 * dxg + 2 * hf;
                const bool va = valid && dxa < 5, vb = valid && dxa + 1 < 5;
                const u32x4 ta = *(wg + ((size_t)((dz * 5 + dy) * 5 + (va ? dxa : 0)) * 2) * 32 + co);
                const u32x4 tb = *(wg + ((size_t)((dz * 5 + dy) * 5 + (vb ? dxa + 1 : 0)) * 2) * 32 + co);
                freg[k] = u32x4{va ? ta[0] : 0u, va ? ta[1] : 0u, vb ? tb[0] : 0u, vb ? tb[1] : 0u};
            } else {
                bool valid;
                const u32x4* src = fsrc(min(tid + k * NT, FUNITS - 1), chunk, valid);
                const u32x4 t = *src;
                const u32x4 z = {0u, 0u, 0u, 0u};
                freg[k] = valid ? t : z;
            }
        }
    };
    auto filter_commit = [&]() {
#pragma unroll
        for (int k = 0; k < FPER; ++k) {
            const int u = tid + k * NT;
            *reinterpret_cast<u32x4*>(u < FUNITS ? fl + (size_t)u * 16 : dump) = freg[k];
        }
    };
    // Step schedule.  One chunk: bricks in order.  Two chunks: bricks in pairs (b0, b1) visited as
    //   (b0, c), (b1, c), (b1, c'), (b0, c')  with c alternating from pair to pair,
    // so the resident filter chunk changes once per pair (every 4th step) instead of at every step; the two bricks'
    // accumulators live in accA / accB and are swapped (32 moves) at the second and fourth step of a pair.
    const bool paired = nch == 2;
    const int nsteps = paired ? (nmine >> 1) * 4 + (nmine & 1) * 2 : nmine * nch;
    auto sched = [&](int s, int& bi, int& ch, bool& first, bool& last, bool& swap) {
        if (!paired) { bi = s / nch; ch = s - bi * nch; first = ch == 0; last = ch == nch - 1; swap = false; return; }
        const int q = s >> 2, r = s & 3, cf = q & 1;
        if (2 * q + 1 >= nmine) { bi = 2 * q; ch = r == 0 ? cf : 1 - cf; first = r == 0; last = r == 1; swap = false; return; }
        const bool second = (r == 1 || r == 2);
        bi = 2 * q + (second ? 1 : 0); ch = r < 2 ? cf : 1 - cf; first = r < 2; last = r >= 2; swap = (r == 1 || r == 3);
    };
    using XH = XTileH<G::IZ, G::IY, G::IX, NT>;
    u32x4 hv[XH::PER];                   // the prefetched tile as it will sit in LDS
    auto tile_issue = [&](int bi, int ch) {
        int b, bz, by, bx;
        brick_origin(bi, b, bz, by, bx);
        XH::template issue_part<0, XH::PER>(hv, reinterpret_cast<const unsigned short*>(a.x0), reinterpret_cast<const unsigned short*>(a.x1),
                                            a.C0, a.C1, ch, b, bz * TZ - 2, by * TY - 2, bx * TX - 2, a.Di, a.Hi, a.Wi, tid);
    };
    auto tile_commit = [&]() {
        if constexpr (IN4) {
            // x-im2col: the 8 bytes (4 modalities) of source voxel ix -> unit (x' = ix - s, plane s >> 1, slot s & 1), s = 0..3
            const int r0 = tid / XH::COLS, col = tid - r0 * XH::COLS;
            const int ix = col >> 1;
            const bool src = (col & 1) == 0 && r0 < XH::RPI;
#pragma unroll
            for (int k = 0; k < XH::PER; ++k) {
                const int row = r0 + k * XH::RPI;
                const bool ok = src && row < XH::ROWS;
                const uint2 v8 = make_uint2(hv[k][0], hv[k][1]);
#pragma unroll
                for (int sft = 0; sft < 4; ++sft) {
                    const int xp = ix - sft;
                    unsigned char* dst = tile + (sft >> 1) * G::PLANE + (row * G::IX + xp) * 16 + (sft & 1) * 8;
                    *reinterpret_cast<uint2*>(ok && xp >= 0 ? dst : dump) = v8;
                }
            }
        } else {
            bf16_tile_commit_h<G, XH, 0, XH::PER>(tile, dump, hv, tid);
        }
    };

    if constexpr (IN4) {
        // units whose source voxel lies beyond the tile are never written (they only ever meet zero weights): make them finite once
        for (int u = tid; u < G::TILE_BYTES / 16; u += NT) reinterpret_cast<u32x4*>(tile)[u] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
    int bi, ch; bool first, last, swp;
    sched(0, bi, ch, first, last, swp);
    filter_issue(ch);
    tile_issue(bi, ch);
    filter_commit();
    tile_commit();
    __syncthreads();

    float bias4[4] = {0.f, 0.f, 0.f, 0.f};       // this lane's four output channels never change: the bias stays in registers
    if (a.bias && 4 * g < a.Cout) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bias4[k] = a.bias[4 * g + k];
    }
    f32x4 accA[4], accB[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) accB[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    VNET_STAMP_DECL;
    for (int step = 0; step < nsteps; ++step) {
        sched(step, bi, ch, first, last, swp);
        const bool more = step + 1 < nsteps;
        int nbi = 0, nchk = ch; bool nf, nl, ns;
        VNET_STAMP_STEP(step);
        VNET_STAMP(0);
        if (more) {
            sched(step + 1, nbi, nchk, nf, nl, ns);
            tile_issue(nbi, nchk);
            __builtin_amdgcn_sched_barrier(0);
        }
        VNET_STAMP(1);
        if (swp) {
#pragma unroll
            for (int m = 0; m < 4; ++m) { const f32x4 t = accA[m]; accA[m] = accB[m]; accB[m] = t; }
        }
        if (first) {
#pragma unroll
            for (int m = 0; m < 4; ++m) accA[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- dz pairs (0,1), (2,3) ----
        VNET_PRIO_ALT(wave >= 4);                  // the younger half first ...
#pragma unroll
        for (int zp = 0; zp < 2; ++zp)
#pragma unroll
            for (int dx = 0; dx < NDX; ++dx) {
                bf16x8 R[8], A[5];
#pragma unroll
                for (int k = 0; k < 8; ++k) R[k] = *reinterpret_cast<const bf16x8*>(bZ + ((2 * zp * G::IY + k) * G::IX + dx * XSTEP) * 16);
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) A[dy] = *reinterpret_cast<const bf16x8*>(fa + ((zp * NDX + dx) * 5 + dy) * 1024);
#pragma unroll
                for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        accA[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[dy], R[m + dy], accA[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);     // keep the next group's 13 fragment reads from being hoisted above this
                                                       // group's MFMAs (the allocator then runs out of registers and spills).
                                                       // With the bf16-source tile (H: 206 VGPRs) an explicit ping-pong of the
                                                       // 13 fragment registers fits (250 VGPRs, no spill) -- measured no faster
                                                       // (0.155-0.172 vs 0.157-0.167 ms at 128^3 16->16): two waves per SIMD
                                                       // already cover the read latency
            }
        VNET_STAMP(2);
                                        // (packing after the first dz pair and letting the second pair's reads hoist: measured 2 % slower)
        // ---- dz = 4: dy pairs (0,1), (2,3) and the single tap dy = 4 ----
        VNET_PRIO_ALT(wave < 4);                   // ... the older half for the last 60 of a step's 260 MFMAs
#pragma unroll
        for (int dx = 0; dx < NDX; ++dx) {
            bf16x8 P[6], S[4], A[3];
#pragma unroll
            for (int k = 0; k < 6; ++k) P[k] = *reinterpret_cast<const bf16x8*>(bY + ((4 * G::IY + k) * G::IX + dx * XSTEP) * 16);
#pragma unroll
            for (int m = 0; m < 4; ++m) S[m] = *reinterpret_cast<const bf16x8*>(b0 + ((4 * G::IY + 4 + m) * G::IX + dx * XSTEP) * 16);
#pragma unroll
            for (int q = 0; q < 3; ++q) A[q] = *reinterpret_cast<const bf16x8*>(fa + (2 * NDX * 5 + dx * 3 + q) * 1024);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                accA[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], P[m], accA[m], 0, 0, 0);
                accA[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], P[m + 2], accA[m], 0, 0, 0);
                accA[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], S[m], accA[m], 0, 0, 0);
            }
        }
        VNET_PRIO_OFF();
        VNET_STAMP(3);
        if (last) {
            // epilogue: lane holds cout 4g..4g+3 of voxel (vz, vy0 + m, x = j)
            int b, bz, by, bx;
            brick_origin(bi, b, bz, by, bx);
            const int oz = bz * TZ + vz, ox = bx * TX + j, co = 4 * g;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            {
                const bool colok = oz < a.Do && ox < a.Wo && co < a.Cout;
                size_t ovs[4]; int cos[4]; bool oks[4]; float e[4][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int oy = by * TY + vy0 + m;
                    oks[m] = colok && oy < a.Ho;
                    ovs[m] = oks[m] ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
                    cos[m] = oks[m] ? co : 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[m][k] = accA[m][k] + bias4[k];
                }
                epilogue_b16_batch<STATS, 4>(a, ovs, cos, oks, e);
                if constexpr (STATS) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (oks[m]) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { s1[k] += e[m][k]; s2[k] += e[m][k] * e[m][k]; }
                        }
                }
            }
            if constexpr (STATS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[k] = row16_sum(s1[k]); s2[k] = row16_sum(s2[k]);
                    if (j == 0) { red[wave * 32 + co + k] = s1[k]; red[wave * 32 + 16 + co + k] = s2[k]; }
                }
            }
        }
        VNET_STAMP(4);
        __syncthreads();                               // every wave is done reading the tile (and the filter chunk)
        VNET_STAMP(5);
        if constexpr (STATS) if (last) stats_row_write<8, 16>(red, a.stats, (size_t)(b_lo + slot + bi * G8), 0, a.Cout, tid);
        if (more) {
            tile_commit();
            // two chunks: the other filter chunk is loaded here, synchronously, once per brick PAIR (every 4th step) -- a
            // register prefetch across the MFMA section does not fit next to the tile prefetch (it spilled)
            if (nchk != ch) { filter_issue(nchk); filter_commit(); }
        }
        VNET_STAMP(6);
        __syncthreads();
        VNET_STAMP(7);
        VNET_STAMP_FLUSH(step, wave, lane, 8);
    }
}

// ------------------------------------------------------------------------------------------
// 32-output-channel blocks on many bricks, bf16 shadows only (round 2).  The generic kernel above reads 1.5 KB of LDS
// fragments per MFMA (two B + one A for two MFMAs), and on this part fragment delivery and MFMA issue add up rather than
// overlap (LDS-active + MFMA-busy cycles = the step time).  Here a wave owns FOUR 32-voxel subtiles = 8
// x-rows adjacent in y, so the B fragment of (subtile m, tap dy) -- rows 2m+dy, 2m+dy+1 -- is fragment F[2m+dy]: per (dz, dx)
// a wave reads 11 row-pair fragments + 5 A fragments for 20 MFMAs = 0.8 KB per MFMA.  Brick 4 x 16 x 16 (8 waves), persistent
// workgroups walk (brick, cout block) items, the next tile is prefetched global -> registers during the MFMAs, and the filter
// streams through TWO LDS plane buffers (one dz plane of one cout block each, 25 KB): one barrier per plane instead of two.
// ------------------------------------------------------------------------------------------
// (bf16 tensors in and out: the fp32-output template path was deleted in round 6)
template <bool STATS = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) conv5_bf16_r32_kernel(ConvArgs a) {
    constexpr int TZ = 4, TY = 16, TX = 16, NT = 512;
    using G = Bf16Geom<TZ, TY, TX>;
    using XH = XTileH<G::IZ, G::IY, G::IX, NT>;
    constexpr int WUNITS = 25 * 64, WPER = (WUNITS + NT - 1) / NT, WBUF = WUNITS * 16 + 16;      // + one dump slot behind each slab
    constexpr int ROWB = G::IX * 16, PLANEB = G::IY * G::IX * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    unsigned char* wbuf = smem + G::TILE_BYTES;                          // two plane buffers
    unsigned char* dump = smem + G::TILE_BYTES + 2 * WBUF + (threadIdx.x & 63) * 16;
    float* red = reinterpret_cast<float*>(smem + G::TILE_BYTES + 2 * WBUF + 64 * 16);            // [8 waves x 2 DPP rows][2 x 32] epilogue statistics

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p32 = lane & 31, half = lane >> 5;
    const int q32 = (p32 >= 16) ? 16 + ((p32 + 12) & 15) : p32;         // second row rotated by 12: conflict-free ds_read_b128 lane groups
    const int vz = wave >> 1, vy0 = (wave & 1) * 8;
    const unsigned char* bb = tile + half * G::PLANE + ((vz * G::IY + vy0 + (q32 >> 4)) * G::IX + (q32 & 15)) * 16;
    const int aoff = (half * 32 + p32) * 16;

    const int ncob = a.CoutP / 32;
    const int nbrick = a.B * a.nbz * a.nby * a.nbx;
    const int nitem = nbrick * ncob;
    const int G8 = gridDim.x >> 3;
    const int per_xcd = (nitem + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int i_lo = xcd * per_xcd, i_hi = min(nitem, i_lo + per_xcd);
    if (i_lo + slot >= i_hi) return;
    const int nmine = (i_hi - i_lo - slot + G8 - 1) / G8;
    const int nch = a.nchunks;
    const int nsteps = nmine * nch;
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp);

    auto step_of = [&](int s, int& brick, int& cob, int& ch) {
        const int it = s / nch; ch = s - it * nch;
        const int gid = i_lo + slot + it * G8;
        brick = gid / ncob; cob = gid - brick * ncob;
    };
    auto origin = [&](int brick, int& b, int& bz, int& by, int& bx) {
        bx = brick % a.nbx; brick /= a.nbx;
        by = brick % a.nby; brick /= a.nby;
        bz = brick % a.nbz; b = brick / a.nbz;
    };
    u32x4 hv[XH::PER];
    auto tile_issue = [&](int brick, int ch) {
        int b, bz, by, bx;
        origin(brick, b, bz, by, bx);
        XH::template issue_part<0, XH::PER>(hv, reinterpret_cast<const unsigned short*>(a.x0), reinterpret_cast<const unsigned short*>(a.x1),
                                            a.C0, a.C1, ch, b, bz * TZ - 2, by * TY - 2, bx * TX - 2, a.Di, a.Hi, a.Wi, tid);
    };
    u32x4 wreg[WPER];
    auto wsrc = [&](int ch, int dz, int cob) { return wg + ((size_t)(ch * 125 + dz * 25) * ncob + cob) * 64; };

    int brick, cob, ch;
    step_of(0, brick, cob, ch);
    bf16_w_issue<1, WPER, NT>(wreg, wsrc(ch, 0, cob), ncob, tid);
    tile_issue(brick, ch);
    bf16_w_commit<1, WPER, NT>(reinterpret_cast<u32x4*>(wbuf), wreg, tid);
    bf16_tile_commit_h<G, XH, 0, XH::PER>(tile, dump, hv, tid);
    __syncthreads();

    f32x16 acc[4];
    int cur = 0;                                   // plane buffer that holds the current dz plane
    VNET_STAMP_DECL;
    for (int step = 0; step < nsteps; ++step) {
        step_of(step, brick, cob, ch);
        const bool more = step + 1 < nsteps;
        int nbrick_ = brick, ncob_ = cob, nch_ = ch;
        VNET_STAMP_STEP(step);
        VNET_STAMP(0);
        if (more) {
            step_of(step + 1, nbrick_, ncob_, nch_);
            tile_issue(nbrick_, nch_);
            __builtin_amdgcn_sched_barrier(0);
        }
        VNET_STAMP(1);
        if (ch == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }
        for (int dz = 0; dz < 5; ++dz) {
            const bool wnext = dz < 4 || more;
            if (wnext) {
                bf16_w_issue<1, WPER, NT>(wreg, dz < 4 ? wsrc(ch, dz + 1, cob) : wsrc(nch_, 0, ncob_), ncob, tid);
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned char* bp = bb + dz * PLANEB;
            const unsigned char* wa = wbuf + cur * WBUF + aoff;
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                bf16x8 F[11], A[5];
#pragma unroll
                for (int p = 0; p < 11; ++p) F[p] = *reinterpret_cast<const bf16x8*>(bp + p * ROWB + dx * 16);
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) A[dy] = *reinterpret_cast<const bf16x8*>(wa + (dy * 5 + dx) * 1024);
#pragma unroll
                for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[dy], F[2 * m + dy], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (wnext) bf16_w_commit<1, WPER, NT>(reinterpret_cast<u32x4*>(wbuf + (cur ^ 1) * WBUF), wreg, tid);
            if (dz < 4) __syncthreads();           // (the last plane's barrier follows the epilogue)
            cur ^= 1;
            VNET_STAMP(2 + dz);
        }
        if (ch == nch - 1) {
            // epilogue: register r of lane = cout co0 + 8*(r/4) + 4*half + r%4 of voxel (subtile m, q32)
            int b, bz, by, bx;
            origin(brick, b, bz, by, bx);
            const int co0 = cob * 32;
            const int oz = bz * TZ + vz, ox = bx * TX + (q32 & 15);
            float s1[4][4], s2[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) s1[g][k] = s2[g][k] = 0.f;
            {
                // bf16 outputs (round 3).  (1) every load of the epilogue -- bias, the other gradient in accumulate mode, the
                // residual of the statistics -- is in flight before the first use; (2) the packed results of channel groups
                // (g, g+1) are exchanged between lanes L and L+32 (v_permlane32_swap), after which a lane holds 8 consecutive
                // channels of its voxel: 16-byte stores, half as many, each half of a 64-byte run instead of a quarter.
                // Stamps before: 6 K (plain) / 11 K (statistics + residual) cycles of a 53-57 K cycle step.
                float bq[4][4];
                int cg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    cg[g] = co0 + g * 8 + half * 4;                               // (a whole 32-cout block: always < Cout)
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.bias) bv = make_float4(a.bias[cg[g]], a.bias[cg[g] + 1], a.bias[cg[g] + 2], a.bias[cg[g] + 3]);
                    bq[g][0] = bv.x; bq[g][1] = bv.y; bq[g][2] = bv.z; bq[g][3] = bv.w;
                }
                size_t ovm[4]; bool vok[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int oy = by * TY + vy0 + 2 * m + (q32 >> 4);
                    vok[m] = oz < a.Do && oy < a.Ho && ox < a.Wo;
                    ovm[m] = vok[m] ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
                }
                // per 16-channel pair of groups: the output tensor it lies in (uniform: y0 / y1 split at a multiple of 16)
                unsigned short* yb[2]; int ycs[2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int c = co0 + pr * 16;
                    const bool iny0 = c < a.Cy0;
                    yb[pr] = iny0 ? reinterpret_cast<unsigned short*>(a.y0) + c : reinterpret_cast<unsigned short*>(a.y1) + (c - a.Cy0);
                    ycs[pr] = iny0 ? a.Cy0 : a.Cy1;
                }
                // one register array for the batched loads: the other gradient (accumulate mode) or else the residual; a launch with
                // both (none in the networks) loads its residual per voxel row below.
                // (a NATIVE vector type: arrays of HIP's uint2 struct are not scalarised -- with two conditional writers this one
                // stayed in scratch in the statistics variant: 32 scratch instructions per brick step, profiles/check_isa.sh; in
                // registers it has to be half as large there, or the statistics variant spills: two voxel rows per batch)
                constexpr int MB = STATS ? 2 : 4;
                const bool res_batched = STATS && a.res && !a.accum;
#pragma unroll
                for (int m0 = 0; m0 < 4; m0 += MB) {
                    u32x2 ld[MB][4];
                    if (a.accum) {
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const unsigned short* sb = a.accsrc ? reinterpret_cast<const unsigned short*>(a.accsrc) + co0 + pr * 16 : yb[pr];
                            const int scs = a.accsrc ? a.Cy0 : ycs[pr];
#pragma unroll
                            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                                for (int gg = 0; gg < 2; ++gg)
                                    ld[mm][2 * pr + gg] = *reinterpret_cast<const u32x2*>(sb + ovm[m0 + mm] * scs + gg * 8 + half * 4);
                        }
                    }
                    if constexpr (STATS) {
                        if (res_batched) {
#pragma unroll
                            for (int mm = 0; mm < MB; ++mm)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    ld[mm][g] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(a.res) + ovm[m0 + mm] * a.Cout + cg[g]);
                        }
                    }
#pragma unroll
                    for (int mm = 0; mm < MB; ++mm) {
                        const int m = m0 + mm;
                        uint2 pk[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float e[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) e[k] = acc[m][g * 4 + k] + bq[g][k];
                            if (a.accum) { e[0] += bf_lo(ld[mm][g].x); e[1] += bf_hi(ld[mm][g].x); e[2] += bf_lo(ld[mm][g].y); e[3] += bf_hi(ld[mm][g].y); }
                            pk[g] = make_uint2(pk_bf16(e[0], e[1]), pk_bf16(e[2], e[3]));
                            if constexpr (STATS) {
                                float v[4] = {bf_lo(pk[g].x), bf_hi(pk[g].x), bf_lo(pk[g].y), bf_hi(pk[g].y)};
                                if (a.res) {
                                    const u32x2 r = res_batched ? ld[mm][g]
                                        : *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(a.res) + ovm[m] * a.Cout + cg[g]);
                                    v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
                                }
                                if (vok[m]) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) { s1[g][k] += v[k]; s2[g][k] += v[k] * v[k]; }
                                }
                            }
                        }
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            // lanes L < 32 keep group 2pr and receive lane L+32's part of it (channels +4..7); lanes L+32 receive
                            // lane L's part of group 2pr+1 and keep their own: 8 consecutive channels each
                            const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * pr].x, pk[2 * pr + 1].x, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * pr].y, pk[2 * pr + 1].y, false, false);
                            const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
                            if (vok[m]) *reinterpret_cast<u32x4*>(yb[pr] + ovm[m] * ycs[pr] + half * 8) = o;
                        }
                    }
                }
            }
            if constexpr (STATS) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s1[g][k] = row16_sum(s1[g][k]);                  // per DPP row (16 lanes); the two rows of a half meet in LDS
                        s2[g][k] = row16_sum(s2[g][k]);
                        if ((lane & 15) == 0) {
                            red[(wave * 2 + (p32 >> 4)) * 64 + g * 8 + half * 4 + k] = s1[g][k];
                            red[(wave * 2 + (p32 >> 4)) * 64 + 32 + g * 8 + half * 4 + k] = s2[g][k];
                        }
                    }
            }
        }
        VNET_STAMP(7);
        __syncthreads();                               // every wave is done with the tile (and the last filter plane; red is complete)
        VNET_STAMP(8);
        if constexpr (STATS) if (ch == nch - 1) stats_row_write<16, 32>(red, a.stats, (size_t)brick, cob * 32, a.Cout, tid);
        if (more) bf16_tile_commit_h<G, XH, 0, XH::PER>(tile, dump, hv, tid);
        VNET_STAMP(9);
        __syncthreads();
        VNET_STAMP(10);
        VNET_STAMP_FLUSH(step, wave, lane, 11);
    }
}

__global__ void pack_bf16_kernel(int mode, const float* __restrict__ w, unsigned short* __restrict__ wp, int T, int I, int O,
                                 int ncob, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
        pack_bf16_elem(mode, w, wp, T, I, O, ncob, idx);
}

void packed_dims_bf16(int mode, int I, int O, int* nchunk, int* ncob) {
    const int K = (mode == VNET_PACK_FWD_BF16) ? I : O, N = (mode == VNET_PACK_FWD_BF16) ? O : I;
    *nchunk = round_up(K, 16) / 16; *ncob = round_up(N, 32) / 32;
}

// ------------------------------------------------------------------------------------------
// bf16-operand filter gradient of the 5x5x5 convolution:  v_mfma_f32_16x16x32_bf16,
//   D[row = cout][col = cin] += A[cout][k = 32 voxels] * B[32 voxels][cin(tap-shifted)]
// Both operands are k-strided in NDHWC memory (k = voxel), which is what the gfx950 LDS transpose read is for:
// the tiles stay [voxel][16 channels] (32-byte rows, bf16) and ds_read_b64_tr_b16 hands lane (i, g) channel i of
// 4 consecutive voxels; two reads = the 8 k-values of lane group g.  A tap shift is a whole number of rows, so it
// is a plain address offset (no alignment constraint) -- per-tap base register + compile-time k-step offset.
// Lane group g of a k-step takes 8 consecutive x; groups 0/1 (one LDS service half) sit on rows y and y+1, whose
// pitch (TX+4 voxels = 640 or 384 bytes) is 128 mod 256, so their 128-byte row quartets use disjoint banks.
// Same work split as the fp32 kernel: 8 waves, wave w owns TW taps x NS cout blocks of one 16-cin chunk.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p, int off) {
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + off + 128));     // rows +4..7
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// (x and dy are bf16 tensors: the fp32-source template path of rounds 2-4 was deleted in round 6)
template <int TZ, int TY, int TX, int NS, int TW>
__device__ __forceinline__ void wgrad5_bf16_body(const WgradArgs& a, const int bid_x, const int bid_y, const int bid_z) {
    using G = TileGeom<5, 1, TZ, TY, TX, 5>;
    using XH = XTileH<G::IZ, G::IY, G::IX, 512>;
    constexpr int NQH = TZ * TY * TX * NS * 2, PERH = (NQH + 511) / 512;     // 16-byte units (voxel, cout block, half) of the dy brick
    constexpr int NV = TZ * TY * TX, T3 = 125;
    constexpr int TXP = TX + 4;                                  // dy row pitch (voxels): same bank argument as the x tile
    constexpr int XBYTES = G::NVOX_IN * 32;
    constexpr int DYPLANE = TZ * TY * TXP * 32;
    static_assert(NV % 32 == 0 && (TX == 16 || TX == 8), "k-steps are 32 voxels: 2 rows of 16 or 4 rows of 8");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;
    unsigned char* dyt = smem + XBYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int split = bid_x;
    const int chunk = bid_y / a.ncob, cob = bid_y - chunk * a.ncob;
    const int co0 = cob * NS * 16;
    const int tap0 = (bid_z * 8 + wave) * TW;

    // lane part of every transpose read: voxel row (group's first voxel + i/4), 8-byte column quad i%4
    const int gx = (TX == 16) ? ((g & 1) * G::IX + (g >> 1) * 8) : g * G::IX;
    const int gd = (TX == 16) ? ((g & 1) * TXP + (g >> 1) * 8) : g * TXP;
    const unsigned char* pt[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        int tap = tap0 + t;
        tap = tap < T3 ? tap : 0;
        const int dx = tap % 5, dy = (tap / 5) % 5, dz = tap / 25;
        pt[t] = xt + (((dz * G::IY + dy) * G::IX + dx) + gx + (i >> 2)) * 32 + (i & 3) * 8;
    }
    const unsigned char* pa = dyt + (gd + (i >> 2)) * 32 + (i & 3) * 8;

    f32x4 acc[TW][NS];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 hx[XH::PER];
    u32x4 hd[PERH];

    auto brick_coords = [&](int brick, int& b, int& bz, int& by, int& bx) {
        bx = brick % a.nbx; brick /= a.nbx;
        by = brick % a.nby; brick /= a.nby;
        bz = brick % a.nbz; b = brick / a.nbz;
    };
    auto issue = [&](int brick) {
        int b, bz, by, bx;
        brick_coords(brick, b, bz, by, bx);
        XH::template issue_part<0, XH::PER>(hx, reinterpret_cast<const unsigned short*>(a.x0), reinterpret_cast<const unsigned short*>(a.x1),
                                            a.C0, a.C1, chunk, b, bz * TZ - 2, by * TY - 2, bx * TX - 2, a.Di, a.Hi, a.Wi, tid);
        const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
#pragma unroll
        for (int k = 0; k < PERH; ++k) {
            const int q = tid + k * 512;
            const int v = q / (NS * 2), cu = q - v * (NS * 2);
            const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
            const int oz = bz * TZ + vz, oy = by * TY + vy, ox = bx * TX + vx;
            const int c = co0 + cu * 8;
            const bool ok = q < NQH && oz < a.Do && oy < a.Ho && ox < a.Wo && c < a.Cout;
            const size_t ov = ok ? ((size_t)(b * a.Do + oz) * a.Ho + oy) * a.Wo + ox : 0;
            hd[k] = load16_or_zero(dyh + ov * a.Cout + (ok ? c : 0), ok);
        }
    };

    if (split < a.nbrick) issue(split);
    for (int brick = split; brick < a.nbrick; brick += a.nsplit) {
        __syncthreads();
        const int r0 = tid / XH::COLS, col = tid - r0 * XH::COLS;
#pragma unroll
        for (int k = 0; k < XH::PER; ++k) {
            const int row = r0 + k * XH::RPI;
            if (r0 < XH::RPI && row < XH::ROWS)
                *reinterpret_cast<u32x4*>(xt + (row * G::IX + (col >> 1)) * 32 + (col & 1) * 16) = hx[k];
        }
#pragma unroll
        for (int k = 0; k < PERH; ++k) {
            const int q = tid + k * 512;
            const int v = q / (NS * 2), cu = q - v * (NS * 2);
            const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
            if (q < NQH) *reinterpret_cast<u32x4*>(dyt + (cu >> 1) * DYPLANE + (((vz * TY + vy) * TXP + vx) * 32) + (cu & 1) * 16) = hd[k];
        }
        __syncthreads();
        if (brick + a.nsplit < a.nbrick) {
            issue(brick + a.nsplit);
            __builtin_amdgcn_sched_barrier(0);
        }
        // K loop in phases: phase p = (k-step p/PH, tap group p%PH).  The reads of phase p+1 are issued before the
        // MFMAs of phase p (ping-pong over TW/PH B fragments; the A fragments change once per k-step).  Short phases
        // keep the fragment registers small: with the next brick's prefetch registers live across this loop, anything
        // above 256 VGPRs spills loop-invariant address parts, and their scratch reloads serialise the prefetch loads
        // (scratch and global loads share vmcnt) -- measured 24 us per brick instead of 2.
        constexpr int NK = NV / 32, PH = (TW >= 16) ? 4 : 2, TH = TW / PH, NP = PH * NK;
        static_assert(TW % PH == 0, "taps are processed in PH groups");
        bf16x8 av[2][NS], bv[2][TH];
        auto fetch = [&](int ph) {                                 // ph is a literal after unrolling
            const int ks = ph / PH, h = ph % PH;
            const int v0 = ks * 32;
            const int vy = (v0 / TX) % TY, vz = v0 / (TX * TY);
            const int xo = ((vz * G::IY + vy) * G::IX) * 32, dyo = ((vz * TY + vy) * TXP) * 32;
            if (h == 0) {
#pragma unroll
                for (int n = 0; n < NS; ++n) av[ks & 1][n] = tr_frag(pa, n * DYPLANE + dyo);
            }
#pragma unroll
            for (int t = 0; t < TH; ++t) bv[ph & 1][t] = tr_frag(pt[h * TH + t], xo);
        };
        fetch(0);
#pragma unroll
        for (int ph = 0; ph < NP; ++ph) {
            if (ph + 1 < NP) fetch(ph + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    acc[(ph % PH) * TH + t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[(ph / PH) & 1][n], bv[ph & 1][t],
                                                                                         acc[(ph % PH) * TH + t][n], 0, 0, 0);
        }
    }
    // lane holds dW[tap][ci = chunk*16 + i][co = co0 + n*16 + 4*g + {0..3}]
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int tap = tap0 + t;
        if (tap >= T3) continue;
        float* dst = a.part + ((size_t)(split * T3 + tap) * a.CinP + chunk * 16 + i) * a.CoutP + co0 + g * 4;
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const f32x4 r = acc[t][n];
            *reinterpret_cast<float4*>(dst + n * 16) = make_float4(r.x, r.y, r.z, r.w);
        }
    }
}

template <int TZ, int TY, int TX, int NS, int TW>
__global__ void __launch_bounds__(512) wgrad5_bf16_kernel(WgradArgs a) {
    wgrad5_bf16_body<TZ, TY, TX, NS, TW>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------------------------------
// Row-reuse filter gradient (round 3; bf16 tensors only).  The kernel above reads one 1 KB B fragment (x, tap-shifted) from LDS per
// MFMA -- with 16 output channels there is a single cout block to share it with -- and on this part fragment delivery and MFMA issue
// add up: 1.06 KB per MFMA, LDS-bound.  Here a k-step is ONE x-row of 32 voxels, so the B fragment of
// (output row y, tap dy) is the fragment of input row y + dy: a wave owns whole dy columns -- three (dz, dx) pairs with all five dy --
// and keeps a sliding window of five row fragments per pair in registers; per output row it reads 3 new B fragments + 1 A
// fragment (+ 1 for its share of the 25th pair, whose five taps go to waves 0-4) for 16 MFMAs: 0.41 KB per MFMA.
//   * brick TZ x 8 x 32 output voxels; x tile (TZ+4) x 12 x 36 voxels x 16 channels and dy tile in LDS as [voxel][16 ch] (32-byte
//     rows, bf16), fragments by ds_read_b64_tr_b16 like the kernel above;
//   * k index of lane group g: voxels {4g..4g+3} and {16+4g..16+4g+3} of the row -- groups 0/1 (one LDS service half) read
//     adjacent 128-byte runs, conflict-free at every tap shift;
//   * D[cout][cin] per tap, one 16-cin chunk x one 16-cout block per workgroup, bricks split over nsplit workgroups, next brick's
//     tiles prefetched global -> registers during the MFMAs; partial slabs + the usual reduce.
// ------------------------------------------------------------------------------------------
// IN4 (round 3): the filter gradient of the network-input conv of a multi-modality net (4 real channels zero-padded to 8, x in
// source 0).  The 16 "input channels" of the tile become j = (x shift sx = 0..3, modality c = 0..3) -- an x-im2col done while the tile
// is committed: a voxel's 8 bytes go to the rows x - sx, bytes 8 sx -- so D[cout][j] of a (dz, dy) holds the taps dx = sx (tile
// offset 0) resp. dx = 4 (offset 4, sx = 0 only): 10 (dz, x offset) pairs instead of 25 (dz, dx) pairs; waves 0-7 own pair w,
// waves 0-1 also pair 8 + w.
template <int TZ, bool IN4 = false>
__device__ __forceinline__ void wgrad5_bf16_rr_body(const WgradArgs& a, const int bid_x, const int bid_y) {
    constexpr int TY = 8, TX = 32, IZ = TZ + 4, IY = TY + 4, IX = TX + 4;
    constexpr int XROWS = IZ * IY, XCOLS = IX * 2, XRPI = 512 / XCOLS, XPER = (XROWS + XRPI - 1) / XRPI;    // x tile: 16-byte units
    constexpr int XBYTES = XROWS * IX * 32;
    constexpr int DUNITS = TZ * TY * TX * 2, DPER = DUNITS / 512;                                            // dy tile units per thread
    static_assert(DUNITS % 512 == 0 && DPER == TZ, "dy tile: one z-slice of 8 x 32 voxels x 2 halves per pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xt = smem;
    unsigned char* dyt = smem + XBYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int split = bid_x;
    const int chunk = bid_y / a.ncob, cob = bid_y - chunk * a.ncob;
    const int co0 = cob * 16;

    // this wave's three (dz, dx) pairs: pair index q = 3 * wave + c  (q = dz * 5 + dx, 0..23); pair 24 = (4, 4) is shared: its tap
    // dy = wave goes to waves 0..4
    const int lane_off = (4 * g + (i >> 2)) * 32 + (i & 3) * 8;
    constexpr int NPAIR = IN4 ? 2 : 3;
    const unsigned char* pb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if constexpr (IN4) {
            const int q = (c == 0) ? wave : (wave < 2 ? 8 + wave : 0), dxg = q / 5, dz = q - dxg * 5;
            pb[c] = xt + ((dz * IY) * IX + 4 * dxg) * 32 + lane_off;
        } else {
            const int q = 3 * wave + c, dz = q / 5, dx = q - dz * 5;
            pb[c] = xt + ((dz * IY) * IX + dx) * 32 + lane_off;
        }
    }
    const bool extra = !IN4 && wave < 5;
    const unsigned char* pe = xt + ((4 * IY + (extra ? wave : 0)) * IX + 4) * 32 + lane_off;
    const unsigned char* pa = dyt + lane_off;

    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 hx[XPER], hd[DPER];
    auto brick_coords = [&](int brick, int& b, int& bz, int& by, int& bx) {
        bx = brick % a.nbx; brick /= a.nbx;
        by = brick % a.nby; brick /= a.nby;
        bz = brick % a.nbz; b = brick / a.nbz;
    };
    // Prefetch addressing: one UNIFORM 64-bit base per brick (source tensor, batch) + a 32-bit element offset per load that depends
    // on the brick (one sample's volume x channels stays below 2^31 elements).  XTileH's loop-invariant 64-bit row offsets were
    // hoisted out of the brick loop by hipcc: 28 registers held for the whole kernel, spilled -- and a scratch reload next to a
    // register prefetch serialises it.  (A 16-channel chunk never straddles the two sources here: dispatch.)
    const int xr0 = tid / XCOLS, xcol = tid - xr0 * XCOLS;
    const int xix = xcol >> 1, xhf = xcol & 1;
    auto issue = [&](int brick) {
        int b, bz, by, bx;
        brick_coords(brick, b, bz, by, bx);
        const int gz0 = bz * TZ - 2, gy0 = by * TY - 2, gx = bx * TX - 2 + xix;
        const int c = chunk * 16;
        const bool first = c < a.C0;
        const int cs = first ? a.C0 : a.C1;
        const unsigned short* src = (first ? reinterpret_cast<const unsigned short*>(a.x0) + c
                                           : reinterpret_cast<const unsigned short*>(a.x1) + (c - a.C0)) + (size_t)b * a.Di * a.Hi * a.Wi * cs;
        const bool colok = xr0 < XRPI && (unsigned)gx < (unsigned)a.Wi && c + xhf * 8 < a.C0 + a.C1;
        constexpr int DIZ = XRPI / IY, DIY = XRPI % IY;
        int row = xr0, jz = xr0 / IY, jy = xr0 - (xr0 / IY) * IY;
#pragma unroll
        for (int k = 0; k < XPER; ++k) {
            const int gz = gz0 + jz, gy = gy0 + jy;
            const bool ok = colok && row < XROWS && (unsigned)gz < (unsigned)a.Di && (unsigned)gy < (unsigned)a.Hi;
            const unsigned off = ok ? (unsigned)(((gz * a.Hi + gy) * a.Wi + gx) * cs + xhf * 8) : 0u;
            hx[k] = load16_or_zero(src + off, ok);
            row += XRPI; jy += DIY; jz += DIZ;
            if (jy >= IY) { jy -= IY; ++jz; }
        }
        // dy tile: thread (voxel of a z-slice, channel half) -> the same (y, x, half) in each of the TZ slices
        const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy) + (size_t)b * a.Do * a.Ho * a.Wo * a.Cout;
        const int v = tid >> 1, hf = tid & 1;
        const int oy = by * TY + (v >> 5), ox = bx * TX + (v & 31), cd = co0 + hf * 8;
        const bool dok = oy < a.Ho && ox < a.Wo && cd < a.Cout;
#pragma unroll
        for (int k = 0; k < DPER; ++k) {              // (DPER == TZ: 512 units per z-slice)
            const int oz = bz * TZ + k;
            const bool ok = dok && oz < a.Do;
            const unsigned off = ok ? (unsigned)(((oz * a.Ho + oy) * a.Wo + ox) * a.Cout + cd) : 0u;
            hd[k] = load16_or_zero(dyh + off, ok);
        }
    };
    auto frag = [&](const unsigned char* p, int off) -> bf16x8 {
        typedef s16x4 __attribute__((address_space(3))) * lp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + off + 512));       // voxels +16..19
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    if (split < a.nbrick) issue(split);
    for (int brick = split; brick < a.nbrick; brick += a.nsplit) {
        __syncthreads();                               // every wave is done reading the previous tiles
        {
            const int r0 = tid / XCOLS, col = tid - r0 * XCOLS;
#pragma unroll
            for (int k = 0; k < XPER; ++k) {
                const int row = r0 + k * XRPI;
                if constexpr (IN4) {
                    if (r0 < XRPI && row < XROWS && (col & 1) == 0) {
                        const uint2 v8 = make_uint2(hx[k][0], hx[k][1]);
#pragma unroll
                        for (int sft = 0; sft < 4; ++sft)
                            if ((col >> 1) - sft >= 0) *reinterpret_cast<uint2*>(xt + (row * IX + (col >> 1) - sft) * 32 + sft * 8) = v8;
                    }
                } else if (r0 < XRPI && row < XROWS)
                    *reinterpret_cast<u32x4*>(xt + (row * IX + (col >> 1)) * 32 + (col & 1) * 16) = hx[k];
            }
#pragma unroll
            for (int k = 0; k < DPER; ++k) *reinterpret_cast<u32x4*>(dyt + (size_t)(tid + k * 512) * 16) = hd[k];
        }
        __syncthreads();
        auto plane = [&](const int z) {
            VNET_PRIO_ALT(((wave >> 2) ^ z) & 1);
            const int zx = z * (IY * IX * 32), zd = z * (TY * TX * 32);
            // one (dz, dx) pair at a time: the window of five row fragments of ONE pair is live (20 registers; all three pairs in
            // lockstep -- A read once per row -- needed 60 and spilled next to the prefetch registers); A is re-read per pair
#pragma unroll
            for (int c = 0; c < NPAIR; ++c) {
                if (IN4 && c == 1 && wave >= 2) continue;                          // (wave-uniform)
                bf16x8 F[5];
#pragma unroll
                for (int r = 0; r < 4; ++r) F[r] = frag(pb[c], zx + r * (IX * 32));
                // the fragments of output row y + 1 are read while row y's MFMAs run (one row ahead, pinned by sched_barrier):
                // the matrix instructions of a row never wait for a read issued in the same row
                bf16x8 An = frag(pa, zd), Fn = frag(pb[c], zx + 4 * (IX * 32)), En = An;
                if (c == 0 && extra) En = frag(pe, zx);                              // (wave-uniform)
#pragma unroll
                for (int y = 0; y < TY; ++y) {
                    const bf16x8 A = An, E = En;
                    F[(y + 4) % 5] = Fn;
                    if (y + 1 < TY) {
                        An = frag(pa, zd + (y + 1) * (TX * 32));
                        Fn = frag(pb[c], zx + (y + 5) * (IX * 32));
                        if (c == 0 && extra) En = frag(pe, zx + (y + 1) * (IX * 32));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy)
                        acc[c * 5 + dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, F[(y + dy) % 5], acc[c * 5 + dy], 0, 0, 0);
                    if (c == 0 && extra) acc[15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, E, acc[15], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // The next brick's loads go out behind the FIRST plane (round 6, as in wgrad5_x3_kernel), not right behind the barrier, where both
        // waves of a SIMD did their address arithmetic at the same time with the matrix pipe idle.  Unconditional: past the end, this brick again.
        plane(0);
        __builtin_amdgcn_sched_barrier(0);
        issue(brick + a.nsplit < a.nbrick ? brick + a.nsplit : brick);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int z = 1; z < TZ; ++z) plane(z);
    }
    VNET_PRIO_OFF();
    if constexpr (IN4) {
        // lane holds D[(dz, dy), x offset 4 dxg][j = i = (sx, c)][co]: the tap dx = 4 dxg + sx of modality c (dx <= 4)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const int c = t / 5;
            if (c == 1 && wave >= 2) continue;
            const int q = (c == 0) ? wave : 8 + wave, dxg = q / 5, dz = q - dxg * 5;
            const int dx = 4 * dxg + (i >> 2);
            if (dx > 4) continue;
            const int tap = (dz * 5 + t % 5) * 5 + dx;
            float* dst = a.part + ((size_t)(split * 125 + tap) * a.CinP + (i & 3)) * a.CoutP + co0 + g * 4;
            const f32x4 r = acc[t];
            *reinterpret_cast<float4*>(dst) = make_float4(r.x, r.y, r.z, r.w);
        }
        return;
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

template <int TZ, bool IN4 = false>
__global__ void __launch_bounds__(512) wgrad5_bf16_rr_kernel(WgradArgs a) {
    wgrad5_bf16_rr_body<TZ, IN4>(a, blockIdx.x, blockIdx.y);
}

template <int TZ, bool IN4 = false>
int launch_wgrad_bf16_rr(const WgradArgs& a, int nsplit, int ncob, hipStream_t st) {
    constexpr int IZ = TZ + 4, IY = 12, IX = 36;
    const size_t lds = (size_t)IZ * IY * IX * 32 + (size_t)TZ * 8 * 32 * 32;
    auto k = wgrad5_bf16_rr_kernel<TZ, IN4>;
    static unsigned long long attr_done = 0;
    if (int ae = ensure_lds(k, lds, attr_done)) return ae;
    dim3 grid(nsplit, (a.CinP / 16) * ncob, 1);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    return (int)hipGetLastError();
}

template <int TZ, int TY, int TX, int NS, int TW>
int launch_wgrad_bf16(const WgradArgs& a, int nsplit, int ncob, int ntg, hipStream_t st) {
    using G = TileGeom<5, 1, TZ, TY, TX, 5>;
    const size_t lds = (size_t)G::NVOX_IN * 32 + (size_t)NS * TZ * TY * (TX + 4) * 32;
    auto k = wgrad5_bf16_kernel<TZ, TY, TX, NS, TW>;
    static unsigned long long attr_done = 0;
    if (int ae = ensure_lds(k, lds, attr_done)) return ae;
    dim3 grid(nsplit, (a.CinP / 16) * ncob, ntg);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    return (int)hipGetLastError();
}

struct Bf16Plan { int nsb, ncobg, nbz, nby, nbx, nsplit, cps, small, nz, half; };

Bf16Plan plan_conv_bf16(int Cin, int Cout, int B, int Do, int Ho, int Wo) {
    Bf16Plan p{};
    const int ncob = round_up(Cout, 32) / 32, nchunks = round_up(Cin, 16) / 16;
    p.small = Wo < 16;
    // few wide bricks (32^3 and below): 4x8x8 bricks with 4-wave workgroups double the workgroup count (+21 % at 32^3
    // 64->64); with many bricks the two shapes measure the same, the wide one stages less halo
#ifdef VNET_PLAN_ENV
    static const int halfmax = getenv("VNET_BF16_HALF_MAX") ? atoi(getenv("VNET_BF16_HALF_MAX")) : 256;
    static const int nsbmin = getenv("VNET_BF16_NSB_MIN") ? atoi(getenv("VNET_BF16_NSB_MIN")) : 256;
#else
    constexpr int halfmax = 256, nsbmin = 256;
#endif
    p.half = !p.small && (long)B * ceil_div(Do, 4) * ceil_div(Ho, 8) * ceil_div(Wo, 16) <= halfmax;
    if (p.small) { p.nbz = ceil_div(Do, 8); p.nby = ceil_div(Ho, 8); p.nbx = ceil_div(Wo, 8); }
    else { p.nbz = ceil_div(Do, 4); p.nby = ceil_div(Ho, 8); p.nbx = ceil_div(Wo, p.half ? 8 : 16); }
    const long nb = (long)B * p.nbz * p.nby * p.nbx;
    p.nsb = (ncob % 2 == 0 && nb * (ncob / 2) >= nsbmin) ? 2 : 1;
    p.ncobg = ncob / p.nsb;
    const long nwg = nb * p.ncobg;
    p.nsplit = 1;
    // split-K target 512 workgroups; the tap (dz) split only below 64: it multiplies the partial slabs by 5, and at 8^3
    // 256->256 (8 x 16 = 128 workgroups) 80 slabs cost 52 us against 26 us with 16 (profiles/ab_plan.sh sweep, round 2:
    // targets 64..1024 x dz-split thresholds 0 / 256 / 1024 on the five deep-level shapes)
#ifdef VNET_PLAN_ENV
    static const int tgt = getenv("VNET_BF16_SPLIT_TARGET") ? atoi(getenv("VNET_BF16_SPLIT_TARGET")) : 512;
    static const int nzmin = getenv("VNET_BF16_NZ_MIN") ? atoi(getenv("VNET_BF16_NZ_MIN")) : 64;
#else
    constexpr int tgt = 512, nzmin = 64;
#endif
    // (nwg == 256, the 32^3 level, stays unsplit: profiles/ab_plan_split.sh -- two workgroups per CU + the reduce pass: 50 vs 42 us)
#ifdef VNET_PLAN_ENV
    static const int splitmax = getenv("VNET_BF16_SPLIT_NWG_MAX") ? atoi(getenv("VNET_BF16_SPLIT_NWG_MAX")) : 255;
#else
    constexpr int splitmax = 255;
#endif
    if (nwg <= splitmax && nchunks > 1) p.nsplit = (int)min((long)nchunks, (long)ceil_div(tgt, (int)nwg));
    p.cps = ceil_div(nchunks, p.nsplit);
    p.nsplit = ceil_div(nchunks, p.cps);
    p.nz = (nwg * p.nsplit < nzmin) ? 5 : 1;
    return p;
}

// 16-cout kernel: exactly the layers that would pad 16 -> 32 cout, vector-aligned channels, >= 256 bricks of 4x8x16
bool conv_bf16_use_c16(int Cin, int Cout, int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) {
    if (Cout > 16 || (Cout & 3) || (C0 & 3) || (C1 & 3) || (Cy0 & 3) || (Cy1 & 3) || W < 16) return false;
    return (long)B * ceil_div(D, 4) * ceil_div(H, 8) * ceil_div(W, 16) >= 256;
}

// ping-pong form (conv_c16pp.h): bf16 tensors in and out, one or two 16-channel chunks, not the x-im2col input layer
// (Cin >= 16: the zero-padded network input -- 8 channels -- keeps the c16 kernel with or without its x-im2col form, so that the
//  number of statistics rows stays a function of the shape alone)
bool conv_bf16_use_c16pp(int Cin, int Cout, int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) {
    if (tuning().bf16_c16pp == 0 || Cin < 16 || (C0 & 7) || (C1 & 7)) return false;
    return conv_bf16_use_c16(Cin, Cout, C0, C1, Cy0, Cy1, B, D, H, W);
}

// row-pair kernel: whole 32-cout blocks, vector-aligned outputs, >= 256 (brick of 4x16x16, cout block) items
bool conv_bf16_use_r32(int Cout, int Cy0, int Cy1, int B, int D, int H, int W) {
#ifdef VNET_PLAN_ENV
    static const int off = getenv("VNET_BF16_R32") ? (atoi(getenv("VNET_BF16_R32")) == 0) : 0;
    if (off) return false;
#endif
    if ((Cout & 31) || (Cy0 & 15) || (Cy1 & 15) || W < 16 || H < 16) return false;      // (16-channel pairs never straddle y0 / y1)
    return (long)B * ceil_div(D, 4) * ceil_div(H, 16) * ceil_div(W, 16) * (Cout / 32) >= 256;
}

template <int TZ, int TY, int TX, int WAVES, bool STATS = false>
int launch_conv_bf16(const ConvArgs& a, const Bf16Plan& p, hipStream_t st) {
    using G = Bf16Geom<TZ, TY, TX>;
    dim3 grid(a.B * p.nbz * p.nby * p.nbx, p.ncobg, p.nsplit * p.nz), block(WAVES * 64);
#define VNET_GO(NSBV)                                                                             \
    {                                                                                             \
        auto k = conv5_bf16_kernel<TZ, TY, TX, NSBV, WAVES, STATS>;                       \
        const size_t lds = (size_t)G::TILE_BYTES + (size_t)25 * NSBV * 1024 + 16 + 64 * 16;       \
        static unsigned long long attr_done = 0;                                                  \
        if (int ae = ensure_lds(k, lds, attr_done)) return ae;                                    \
        hipLaunchKernelGGL(k, grid, block, lds, st, a);                                           \
    }
    if (p.nsb == 2) VNET_GO(2) else VNET_GO(1)
#undef VNET_GO
    return (int)hipGetLastError();
}

}  // namespace

namespace {
struct WgradPlan { int ns, tw, ncob, ntg, nbz, nby, nbx, nbrick, nsplit, small; };

WgradPlan plan_wgrad(int ks, int kx, int stride, int Cin, int Cout, int B, int Do, int Ho, int Wo, bool bf16 = false) {
    WgradPlan p{};
    const int CoutP = round_up(Cout, 16);
    p.ns = pick_ns(CoutP);
    // 5^3: at most two cout blocks per workgroup.  Four would halve the staging per MFMA, but the layers wide enough for it are
    // the deep ones with few bricks, where twice the (chunk x cout-block) workgroups means half the filter slabs to reduce
    // and a pipeline fill amortised over twice the bricks: +2.5..5 % measured (and the bf16 kernel's 4-block variant does
    // not fit 256 VGPRs next to its prefetch registers)
    if (ks == 5 && p.ns == 4) p.ns = 2;
    p.ncob = CoutP / (16 * p.ns);
    p.small = Wo < 16;
    const int T3 = ks * ks * kx;
    if (ks == 5) {
        p.tw = kx == 1 ? 4 : 16 / p.ns;
        if (p.small) { p.nbz = ceil_div(Do, 4); p.nby = ceil_div(Ho, 8); p.nbx = ceil_div(Wo, 8); }
        else { p.nbz = ceil_div(Do, 4); p.nby = ceil_div(Ho, 4); p.nbx = ceil_div(Wo, 16); }
    } else {  // ks == 2, stride 2: out brick 2x4x16 / 2x8x8
        p.tw = 1;
        if (p.small) { p.nbz = ceil_div(Do, 2); p.nby = ceil_div(Ho, 8); p.nbx = ceil_div(Wo, 8); }
        else { p.nbz = ceil_div(Do, 2); p.nby = ceil_div(Ho, 4); p.nbx = ceil_div(Wo, 16); }
    }
    p.ntg = ceil_div(T3, 8 * p.tw);
    p.nbrick = B * p.nbz * p.nby * p.nbx;
    const int base = (round_up(Cin, 16) / 16) * p.ncob * p.ntg;
#ifdef VNET_PLAN_ENV
    static const int wtgt = getenv("VNET_WGRAD_TARGET") ? atoi(getenv("VNET_WGRAD_TARGET")) : 256;
#else
    constexpr int wtgt = 256;
#endif
    p.nsplit = max(1, min(p.nbrick, ceil_div(wtgt, base)));
    return p;
}

template <int KS, int STRIDE, int TZ, int TY, int TX, int NS, int TW, int KX = KS, bool IO16 = false>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
    using G = TileGeom<KS, STRIDE, TZ, TY, TX, KX>;
    const size_t lds = ((size_t)G::LDS_FLOATS + (size_t)TZ * TY * TX * NS * 16) * 4;
    auto k = wgrad_kernel<KS, STRIDE, TZ, TY, TX, NS, TW, KX, IO16>;
    static unsigned long long attr_done = 0;
    if (int ae = ensure_lds(k, lds, attr_done)) return ae;
    dim3 grid(p.nsplit, (a.CinP / 16) * p.ncob, p.ntg);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace
#include "conv_c16pp.h"
namespace {
// kernel choice of the bf16-storage 5^3 convolution (bf16 tensors in and out); -1 = launched, nothing to reduce
// (a template so that only the translation unit that calls it instantiates the kernels)
template <int UNIT = 0>
int conv_fwd_bf16_go(ConvArgs& a, const Bf16Plan& p, int nslab, int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W, hipStream_t st) {
    if (conv_bf16_use_c16(a.Cin, a.Cout, C0, C1, Cy0, Cy1, B, D, H, W)) {
        // 16 output channels at a size with enough bricks for one persistent workgroup per CU: no padding to 32 cout
        using GC = Bf16Geom<4, 8, 16>;
        a.nbz = ceil_div(D, 4); a.nby = ceil_div(H, 8); a.nbx = ceil_div(W, 16);
        const size_t lds = (size_t)GC::TILE_BYTES + 65 * 1024 + 64 * 16 + 8 * 32 * 4;
        {
            if (!a.in4 && conv_bf16_use_c16pp(a.Cin, a.Cout, C0, C1, Cy0, Cy1, B, D, H, W)) {
                // filter out of LDS, two 4-wave workgroups per CU (conv_c16pp.h): the same 4x8x16 bricks
                const int grid = 2 * (device_cus() / 8) * 8;
                if (a.stats) {
                    auto k = conv5_bf16_c16pp_kernel<true>;
                    static unsigned long long attr_done = 0;
                    if (int ae = ensure_lds(k, PP_LDS, attr_done)) return ae;
                    hipLaunchKernelGGL(k, dim3(grid), dim3(256), PP_LDS, st, a);
                } else {
                    auto k = conv5_bf16_c16pp_kernel<false>;
                    static unsigned long long attr_done = 0;
                    if (int ae = ensure_lds(k, PP_LDS, attr_done)) return ae;
                    hipLaunchKernelGGL(k, dim3(grid), dim3(256), PP_LDS, st, a);
                }
                VNET_LAUNCH_CHECK();
                return -1;
            }
            if (a.in4) {              // the multi-modality network input: x-im2col in LDS, 2.5x fewer MFMAs
                if (a.stats) {
                    auto k = conv5_bf16_c16_kernel<4, 8, 16, true, true>;
                    static unsigned long long attr_done = 0;
                    if (int ae = ensure_lds(k, lds, attr_done)) return ae;
                    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, st, a);
                } else {
                    auto k = conv5_bf16_c16_kernel<4, 8, 16, false, true>;
                    static unsigned long long attr_done = 0;
                    if (int ae = ensure_lds(k, lds, attr_done)) return ae;
                    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, st, a);
                }
                VNET_LAUNCH_CHECK();
                return -1;
            }
        }
        if (a.stats) {
            auto k = conv5_bf16_c16_kernel<4, 8, 16, true>;
            static unsigned long long attr_done = 0;
            if (int ae = ensure_lds(k, lds, attr_done)) return ae;
            hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, st, a);
        } else {
            auto k = conv5_bf16_c16_kernel<4, 8, 16, false>;
            static unsigned long long attr_done = 0;
            if (int ae = ensure_lds(k, lds, attr_done)) return ae;
            hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, st, a);
        }
        VNET_LAUNCH_CHECK();
        return -1;            // done, no reduce
    }
    {
        if (conv_bf16_use_r32(a.Cout, Cy0, Cy1, B, D, H, W) && nslab == 1) {
            // 32-cout blocks, many bricks, bf16 shadows: the row-pair kernel (11 B + 5 A fragments per 20 MFMAs)
            using GR = Bf16Geom<4, 16, 16>;
            a.nbz = ceil_div(D, 4); a.nby = ceil_div(H, 16); a.nbx = ceil_div(W, 16);
            const size_t lds = (size_t)GR::TILE_BYTES + 2 * (25 * 1024 + 16) + 64 * 16 + 16 * 64 * 4;
            if (a.stats) {
                auto k = conv5_bf16_r32_kernel<true>;
                static unsigned long long attr_done = 0;
                if (int ae = ensure_lds(k, lds, attr_done)) return ae;
                hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, st, a);
            } else {
                auto k = conv5_bf16_r32_kernel<false>;
                static unsigned long long attr_done = 0;
                if (int ae = ensure_lds(k, lds, attr_done)) return ae;
                hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, st, a);
            }
            VNET_LAUNCH_CHECK();
            return -1;
        }
    }
    return (a.stats && nslab == 1)
         ? (p.small ? launch_conv_bf16<8, 8, 8, 8, true>(a, p, st)
            : p.half ? launch_conv_bf16<4, 8, 8, 4, true>(a, p, st) : launch_conv_bf16<4, 8, 16, 8, true>(a, p, st))
         : (p.small ? launch_conv_bf16<8, 8, 8, 8, false>(a, p, st)
            : p.half ? launch_conv_bf16<4, 8, 8, 4, false>(a, p, st) : launch_conv_bf16<4, 8, 16, 8, false>(a, p, st));
}

}  // namespace

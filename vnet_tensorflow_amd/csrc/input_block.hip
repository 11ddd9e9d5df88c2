// input_block.hip -- the single-modality input block of the V-Net without the 16x redundant work.
//
// Reference networks.py:254-259: a 1-channel image is tf.tile'd to num_channels and batch-normalised, so every
// channel of the first 5x5x5 convolution's input (networks.py:316, encoder level 1 conv_1) is an affine function
// of the SAME image:  x_c = a_c * img + b_c  with  a_c = gamma_c * invstd,  b_c = beta_c - mean * a_c.  Hence
//     conv(x, w)[v][o] = sum_tap (sum_c w[tap][c][o] a_c) img[v+tap] + sum_{tap: v+tap inside} (sum_c w[tap][c][o] b_c)
// i.e. a 2-channel convolution of (img, inside-indicator) with folded filters.  The x taps of those two channels
// are im2col'ed into 10 of 16 "virtual channels" so the MFMA kernel runs 25 taps x 16 channels (conv 5x5x1)
// instead of 125 x 16: 5x fewer matrix instructions forward, the same for the filter gradient, and the
// backward-data convolution disappears altogether because only its per-channel reductions are needed
// (d gamma_c, d beta_c of the input batch-norm), which follow from the 2-channel filter gradient G:
//     dw[tap][c][o]   = a_c G1[tap][o] + b_c G2[tap][o]
//     d beta_c  (+)=  sum_{tap,o} w[tap][c][o] G2[tap][o]
//     d gamma_c (+)=  sum_{tap,o} w[tap][c][o] invstd_c (G1[tap][o] - mean_c G2[tap][o])
// Exact in exact arithmetic; in fp32 it differs from the tiled computation only by summation order.
#include "common.h"

namespace {

// xv[v][2*dx + 0] = img[z][y][x+dx-2] (0 outside), xv[v][2*dx + 1] = 1 if x+dx-2 inside else 0, channels 10..15 = 0
__global__ void __launch_bounds__(256) im2col_x_kernel(const float* __restrict__ img, float* __restrict__ xv, size_t nvox, int W) {
    const size_t nq = nvox * 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        const size_t v = q >> 2;
        const int cq = (int)(q & 3);
        const int x = (int)(v % W);
        float e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = cq * 4 + k, dx = ch >> 1;
            const int xx = x + dx - 2;
            const bool in = ch < 10 && xx >= 0 && xx < W;
            e[k] = !in ? 0.f : ((ch & 1) ? 1.f : img[v + dx - 2]);
        }
        reinterpret_cast<float4*>(xv)[q] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// wv[t25][vch][o]: vch = 2*dx + {0: folded with a_c, 1: folded with b_c}
__global__ void __launch_bounds__(256) fold_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ wv, int C, int O) {
    const int total = 25 * 16 * O;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int o = idx % O, vch = (idx / O) % 16, t25 = idx / (16 * O);
        float s = 0.f;
        if (vch < 10) {
            const int dx = vch >> 1, tap = t25 * 5 + dx;
            for (int c = 0; c < C; ++c) {
                const float a = gamma[c] * invstd[c];
                const float coef = (vch & 1) ? (beta[c] - mean[c] * a) : a;
                s += w[((size_t)tap * C + c) * O + o] * coef;
            }
        }
        wv[idx] = s;
    }
}

// one workgroup per input channel c: dw[:, c, :] and the conv-path parts of d gamma_c / d beta_c
__global__ void __launch_bounds__(256) input_grads_kernel(const float* __restrict__ G, const float* __restrict__ w,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          float* __restrict__ dw, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int C, int O, int accumulate) {
    __shared__ double sh[2][4];
    const int c = blockIdx.x;
    const float a = gamma[c] * invstd[c], b = beta[c] - mean[c] * a;
    double sg = 0.0, sb = 0.0;
    for (int idx = threadIdx.x; idx < 125 * O; idx += blockDim.x) {
        const int o = idx % O, tap = idx / O;
        const int t25 = tap / 5, dx = tap - t25 * 5;
        const float g1 = G[((size_t)t25 * 16 + 2 * dx) * O + o], g2 = G[((size_t)t25 * 16 + 2 * dx + 1) * O + o];
        const size_t wi = ((size_t)tap * C + c) * O + o;
        const float wt = w[wi];
        dw[wi] = a * g1 + b * g2;
        sg += (double)wt * (double)(invstd[c] * (g1 - mean[c] * g2));
        sb += (double)wt * (double)g2;
    }
    sg = wave_sum_d(sg); sb = wave_sum_d(sb);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = sg; sh[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tg = (float)(sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        const float tb = (float)(sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
        dgamma[c] = accumulate ? dgamma[c] + tg : tg;
        dbeta[c] = accumulate ? dbeta[c] + tb : tb;
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Round 6: the input block WITHOUT the matrix cores and without the x-im2col tensor.
// The folded form above is a 2-channel convolution (image, inside-indicator): 2 x 125 x O multiply-adds per voxel.  As a 5x5x1
// convolution over 16 virtual channels it kept the MFMA pipe busy with 26.8 GF at 128^3 (10 of 16 channels used, fp32 MFMA at 1/16 of
// the bf16 rate: 0.30 ms forward + 0.29 ms filter gradient per step, plus a 134 MB im2col tensor written once and read twice).  Here:
//   * the IMAGE channel -- 125 x O multiply-adds per voxel, 4.2 GF -- runs on the VECTOR pipe with packed fp32 FMAs (v_pk_fma_f32:
//     157 TF/s, the matrix pipe's own fp32 rate) straight from the 1-channel image (8 MB); every product an exact fp32 FMA;
//   * the INDICATOR channel needs no multiplications at all: ind(v + tap) = in_z(z + dz) in_y(y + dy) in_x(x + dx) is separable, so
//     forward its x taps are pre-summed per x CLASS of the voxel ((min(x, 2), min(W - 1 - x, 2)): 9 classes, vnet_input_conv_fold_border)
//     -- a constant per class where the brick's z / y halo is inside the volume (88 % of the bricks at 128^3), 25 adds per voxel on the
//     z / y faces -- and backward it is box sums of dy: row sums minus the edge columns.
// Forward: a workgroup = 4 waves = a brick of 4 (z) x 4 (y) x 64 (x) voxels; wave = y row, lane = x, a thread owns the 4 voxels of its
// z column (4 x O accumulators); image brick + halo (8 x 8 x 68 floats) in LDS; per tap the O folded weights arrive as SCALAR operands
// (s_load, uniform address), four LDS reads (one per z) feed 4 x O/2 v_pk_fma_f32.  Epilogue: indicator term, bias, y stored as 64
// contiguous bytes per voxel, batch-norm statistics of y (+ residual) as one row per brick.
// ------------------------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
// sum over the 16 lanes of a DPP row, every lane gets it (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

constexpr int IC_TZ = 4, IC_TY = 4, IC_TX = 64;
constexpr int IC_IZ = IC_TZ + 4, IC_IY = IC_TY + 4, IC_IX = IC_TX + 4;
constexpr int IC_TILE = IC_IZ * IC_IY * IC_IX;          // 4352 floats

// 16 bytes of zeros in device memory: masked-off lanes load from here (a select on the ADDRESS keeps the load unconditional).
// Explicit global address space on both sides of the select: a plain pointer select falls back to flat_load, which also counts on
// lgkmcnt and is waited for by the next LDS access.
__device__ __attribute__((aligned(16))) const float ic_zero4[4] = {0.f, 0.f, 0.f, 0.f};
typedef const __attribute__((address_space(1))) float* ic_gf_t;
typedef const __attribute__((address_space(1))) f32x4* ic_gf4_t;      // (the ext-vector type: HIP's float4 struct behind an address-space pointer loads flat)
__device__ __forceinline__ float ic_ld1(const float* p, bool ok) { ic_gf_t q = ok ? (ic_gf_t)p : (ic_gf_t)ic_zero4; return *q; }
__device__ __forceinline__ float4 ic_ld4(const float* p, bool ok) { ic_gf4_t q = ok ? (ic_gf4_t)p : (ic_gf4_t)ic_zero4; const f32x4 v = *q; return make_float4(v[0], v[1], v[2], v[3]); }

// The filter gradient walks bricks of IW_TZ = 2 planes (512 voxels): image tile 6 x 8 x 68 floats + dy brick 32 KB + the row sums =
// 49 KB of LDS, so three workgroups share a CU and one's loads / commits / indicator sums run under another's MFMAs (with the forward
// kernel's 4-plane brick -- 87 KB, one workgroup per CU -- every phase of a brick was exposed).
constexpr int IW_TZ = 2, IW_IZ = IW_TZ + 4;
constexpr int IW_TILE = IW_IZ * IC_IY * IC_IX;          // 3264 floats
constexpr int IW_NVB = IW_TZ * IC_TY * IC_TX;           // 512 voxels
constexpr int IW_NI = (IW_TILE + 255) / 256;            // 13 image loads per thread
constexpr int IW_ND = IW_NVB * 4 / 256;                 // 8 float4 of dy per thread

// the loads of one brick of the filter gradient, global -> registers
template <int O>
__device__ __forceinline__ void ic_issue_brick(float4 (&dv)[IW_ND], float (&iv)[IW_NI], const float* __restrict__ img, const float* __restrict__ dy,
                                               int brick, int nbz, int nby, int nbx, int D, int H, int W, int tid) {
    const int bx = brick % nbx; brick /= nbx;
    const int by = brick % nby; brick /= nby;
    const int bz = brick % nbz; const int b = brick / nbz;
    const int gz0 = bz * IW_TZ - 2, gy0 = by * IC_TY - 2, gx0 = bx * IC_TX - 2;
    const float* src = img + (size_t)b * D * H * W;
#pragma unroll
    for (int k = 0; k < IW_ND; ++k) {
        const int q = tid + 256 * k;
        const int v = q >> 2, cq = q & 3;
        const int vx = v % IC_TX, vy = (v / IC_TX) % IC_TY, vz = v / (IC_TX * IC_TY);
        const int oz = bz * IW_TZ + vz, oy = by * IC_TY + vy, ox = bx * IC_TX + vx;
        const bool ok = 4 * cq < O && oz < D && oy < H && ox < W;
        dv[k] = ic_ld4(dy + ((((size_t)b * D + oz) * H + oy) * W + ox) * O + 4 * cq, ok);
    }
#pragma unroll
    for (int k = 0; k < IW_NI; ++k) {
        const int q = min(tid + 256 * k, IW_TILE - 1);
        const int ix = q % IC_IX, r = q / IC_IX, iy = r % IC_IY, iz = r / IC_IY;
        const int gz = gz0 + iz, gy = gy0 + iy, gx = gx0 + ix;
        const bool in = (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        iv[k] = ic_ld1(src + ((size_t)gz * H + gy) * W + gx, in);
    }
}

// image brick + halo -> LDS: all 17 loads of a thread in flight, then the stores
__device__ __forceinline__ void ic_load_tile(float* __restrict__ timg, const float* __restrict__ src, int gz0, int gy0, int gx0, int D, int H, int W, int tid) {
    float v[IC_TILE / 256];
#pragma unroll
    for (int k = 0; k < IC_TILE / 256; ++k) {
        const int q = tid + 256 * k;
        const int ix = q % IC_IX, r = q / IC_IX, iy = r % IC_IY, iz = r / IC_IY;
        const int gz = gz0 + iz, gy = gy0 + iy, gx = gx0 + ix;
        const bool in = (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        v[k] = ic_ld1(src + ((size_t)gz * H + gy) * W + gx, in);
    }
#pragma unroll
    for (int k = 0; k < IC_TILE / 256; ++k) timg[tid + 256 * k] = v[k];
}

// x class of a voxel: which of the five x taps stay inside the volume (dx valid <=> 2 - lo <= dx <= 2 + hi)
__device__ __forceinline__ int ic_xclass(int ox, int W) { return min(ox, 2) * 3 + min(W - 1 - ox, 2); }

// wbc[cls][t25][o] = sum over the valid dx of class cls of wv[t25][2 dx + 1][o];  cbc[cls][o] = sum_t25 wbc[cls][t25][o]
// (one workgroup per class; the sums of the 25 rows go through LDS: a thread that walks all 125 taps alone costs 19 us of load latency)
__global__ void __launch_bounds__(256) input_fold_border_kernel(const float* __restrict__ wv, int O, float* __restrict__ wbc, float* __restrict__ cbc) {
    __shared__ float row[25 * 16];
    const int cls = blockIdx.x, lo = cls / 3, hi = cls - lo * 3;
    for (int q = threadIdx.x; q < 25 * O; q += 256) {
        const int o = q % O, t25 = q / O;
        float t[5];
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) t[dx] = wv[(t25 * 16 + 2 * dx + 1) * O + o];
        float s = 0.f;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) if (dx >= 2 - lo && dx <= 2 + hi) s += t[dx];
        wbc[(size_t)cls * 25 * O + q] = s;
        row[q] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < O) {
        float s = 0.f;
        for (int t25 = 0; t25 < 25; ++t25) s += row[t25 * O + threadIdx.x];
        cbc[cls * O + threadIdx.x] = s;
    }
}

struct ICArgs {
    const float* img; const float* wv; const float* wbc; const float* cbc; const float* bias; const float* res; float* y; float* stats;
    int B, D, H, W, nbz, nby, nbx;
};

template <int O>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) input_conv_direct_kernel(ICArgs a) {
    static_assert(O == 8 || O == 16, "output channels: 8 or 16");
    __shared__ float timg[IC_TILE];
    __shared__ float red[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    int brick = blockIdx.x;
    const int bx = brick % a.nbx; brick /= a.nbx;
    const int by = brick % a.nby; brick /= a.nby;
    const int bz = brick % a.nbz; const int b = brick / a.nbz;
    const int gz0 = bz * IC_TZ - 2, gy0 = by * IC_TY - 2, gx0 = bx * IC_TX - 2;
    const bool inner_zy = gz0 >= 0 && gz0 + IC_IZ <= a.D && gy0 >= 0 && gy0 + IC_IY <= a.H;      // z / y halo inside the volume
    // A fragments of the 32 K-steps (k = tap 4 s + kk, row m = output channel i): the whole folded filter of the image channel, 32
    // registers for the life of the workgroup
    float af[32];
    int toff[32];                                     // tile offset of this lane's tap in step s
#pragma unroll
    for (int st = 0; st < 32; ++st) {
        const int tap = 4 * st + kk;
        const int t25 = tap / 5, dx = tap - t25 * 5, dz = t25 / 5, dy = t25 - dz * 5;
        const bool ok = tap < 125 && i < O;
        af[st] = ok ? a.wv[(t25 * 16 + 2 * dx) * O + i] : 0.f;
        toff[st] = tap < 125 ? (dz * IC_IY + dy) * IC_IX + dx : 0;
    }
    ic_load_tile(timg, a.img + (size_t)b * a.D * a.H * a.W, gz0, gy0, gx0, a.D, a.H, a.W, tid);
    __syncthreads();

    // D[16 o][16 voxels] += A[o][k = 4 taps] B[4 taps][16 voxels]: a subtile = 16 consecutive x of one (z, y) row; the four z of a column
    // of subtiles run interleaved (four independent accumulators per A fragment)
    const int oy = by * IC_TY + wave;
    const int co = 4 * kk;                            // D layout: lane (column n = voxel i, kk) holds rows 4 kk .. 4 kk + 3 = channels co ..
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && co < O) { const float4 bb = *reinterpret_cast<const float4*>(a.bias + co); bias4[0] = bb.x; bias4[1] = bb.y; bias4[2] = bb.z; bias4[3] = bb.w; }
#pragma unroll 1
    for (int xq = 0; xq < IC_TX / 16; ++xq) {
        const float* tb = timg + wave * IC_IX + xq * 16 + i;
        f32x4 acc[IC_TZ];
#pragma unroll
        for (int z = 0; z < IC_TZ; ++z) acc[z] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the residual of the statistics is fetched under the MFMAs of this column (unconditional: masked lanes read a zero line)
        float4 rr[IC_TZ];
        {
            const int ox_ = bx * IC_TX + xq * 16 + i;
#pragma unroll
            for (int z = 0; z < IC_TZ; ++z) {
                const int oz = bz * IC_TZ + z;
                const bool ok = a.stats && a.res && oz < a.D && oy < a.H && ox_ < a.W && co < O;
                rr[z] = ic_ld4(a.res + ((((size_t)b * a.D + oz) * a.H + oy) * a.W + ox_) * O + co, ok);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#if !defined(IC_ABL) || IC_ABL != 1
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            float bv[IC_TZ];
#pragma unroll
            for (int z = 0; z < IC_TZ; ++z) bv[z] = tb[toff[st] + z * IC_IY * IC_IX];
#pragma unroll
            for (int z = 0; z < IC_TZ; ++z) acc[z] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st], bv[z], acc[z], 0, 0, 0);
        }
#else
        acc[0][0] = tb[toff[xq]] * af[xq];           // (timing ablation: no MFMA loop; results are wrong)
#endif
        const int ox = bx * IC_TX + xq * 16 + i;
        const bool col_ok = oy < a.H && ox < a.W && co < O;
        const int cls = ic_xclass(min(ox, a.W - 1), a.W);
        // indicator channel: per x class a constant (z / y halo inside) or the sum over the (dz, dy) whose row lies inside the volume
        if (co < O) {
            if (inner_zy) {
                const float4 c = *reinterpret_cast<const float4*>(a.cbc + (size_t)cls * O + co);
#pragma unroll
                for (int z = 0; z < IC_TZ; ++z) { acc[z][0] += c.x; acc[z][1] += c.y; acc[z][2] += c.z; acc[z][3] += c.w; }
            } else {
#pragma unroll 1
                for (int t25 = 0; t25 < 25; ++t25) {
                    const int dz = t25 / 5, dy = t25 - dz * 5;
                    if ((unsigned)(oy + dy - 2) >= (unsigned)a.H) continue;                 // (wave-uniform)
                    const float4 c = *reinterpret_cast<const float4*>(a.wbc + ((size_t)cls * 25 + t25) * O + co);
#pragma unroll
                    for (int z = 0; z < IC_TZ; ++z) {
                        if ((unsigned)(bz * IC_TZ + z + dz - 2) < (unsigned)a.D) { acc[z][0] += c.x; acc[z][1] += c.y; acc[z][2] += c.z; acc[z][3] += c.w; }
                    }
                }
            }
        }
#pragma unroll
        for (int z = 0; z < IC_TZ; ++z) {
            const int oz = bz * IC_TZ + z;
#if defined(IC_ABL) && IC_ABL == 2
            if (oz < a.D && col_ok && acc[z][0] == 12345.678f) {      // (timing ablation: no stores / residual loads)
#else
            if (oz < a.D && col_ok) {
#endif
                const size_t ov = (((size_t)b * a.D + oz) * a.H + oy) * a.W + ox;
                const float e[4] = {acc[z][0] + bias4[0], acc[z][1] + bias4[1], acc[z][2] + bias4[2], acc[z][3] + bias4[3]};
                *reinterpret_cast<float4*>(a.y + ov * O + co) = make_float4(e[0], e[1], e[2], e[3]);
#if defined(IC_ABL) && IC_ABL == 3
                if (false) {
#else
                if (a.stats) {
#endif
                    const float vv[4] = {e[0] + rr[z].x, e[1] + rr[z].y, e[2] + rr[z].z, e[3] + rr[z].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s1[k] += vv[k]; s2[k] += vv[k] * vv[k]; }
                }
            }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t1 = row16_sum(s1[k]), t2 = row16_sum(s2[k]);           // over the 16 voxels of a lane row: channels co + k
            if (i == 0 && co < O) { red[wave * 2 * O + co + k] = t1; red[wave * 2 * O + O + co + k] = t2; }
        }
        __syncthreads();
        if (tid < 2 * O) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[w * 2 * O + tid];
            a.stats[(size_t)blockIdx.x * 2 * O + tid] = t;
        }
    }
}

// Filter gradient of the folded 2-channel convolution:  G[t25][2 dx][o] = sum_v img[v + tap] dy[v][o] (packed fp32 FMAs),
// G[t25][2 dx + 1][o] = sum_{v: v + tap inside} dy[v][o] (box sums of dy: no multiplications) -- the layout vnet_input_conv_grads
// reads; virtual channels 10..15 = 0.  Persistent workgroups of 256 threads walk bricks of 4 x 4 x 64 voxels; image tile and the dy
// brick (1024 voxels x O) sit in LDS.
//   image channel: thread = (tap quad tq of 32, voxel slice s of 8): 4 taps x O accumulators; per voxel of its slice: O/4 broadcast
//     reads of dy, 4 image reads, 4 x O/2 packed FMAs; at the end the 8 slices meet in LDS;
//   indicator channel: per brick the 16 (z, y) rows' sums of dy over x, with the edge columns (x = 0, 1, W - 2, W - 1) taken out per
//     dx -> T[row][dx][o]; tap (dz, dy, dx) then collects the rows whose shifted row lies inside the volume (all 16 when the brick's
//     z / y halo is inside: one brick sum per dx).  Every output (tap, o) has ONE owner thread: 8 accumulators per thread.
// One partial slab [25][16][O] per workgroup; the slabs are summed by input_wgrad_reduce_kernel.
template <int O>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) input_wgrad_direct_kernel(const float* __restrict__ img, const float* __restrict__ dy, float* __restrict__ part,
                                                                  int B, int D, int H, int W, int nbz, int nby, int nbx) {
    static_assert(O == 8 || O == 16, "output channels: 8 or 16");
    constexpr int NVB = IW_NVB;                       // 512 voxels per brick
    constexpr int NROW = IW_TZ * IC_TY;              // 8 (z, y) rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* timg = smem;                              // IW_TILE
    float* tT = smem + IW_TILE;                      // [NROW + 1][5][O]: T per row, and the brick sums in the last slot
    float* tdy = smem + IW_TILE + (NROW + 1) * 5 * 16;   // NVB * 16 (always 16 columns: the B fragment reads column i; O = 8: columns 8..15 zero)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kk = lane >> 4;
    // image channel on the matrix cores: D[16 taps][16 o] += A[tap][k = 4 voxels] B[4 voxels][o], eight M tiles = taps 16 m + i; the
    // A fragment of a lane is the image at its voxel kk shifted by ITS tap: one LDS read at a per-lane offset (im2col by addressing)
    int toff[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int tap = min(16 * m + i, 124);
        const int dz = tap / 25, dyy = (tap / 5) % 5, dx = tap % 5;
        toff[m] = (dz * IC_IY + dyy) * IC_IX + dx + kk;
    }
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float g2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g2[j] = 0.f;
    const int nbrick = B * nbz * nby * nbx;
    // the next brick's image tile (17 floats) and dy brick (16 float4) travel global -> registers under this brick's MFMAs
    float4 dv[IW_ND];
    float iv[IW_NI];
    if ((int)blockIdx.x < nbrick) ic_issue_brick<O>(dv, iv, img, dy, blockIdx.x, nbz, nby, nbx, D, H, W, tid);
    for (int brick0 = blockIdx.x; brick0 < nbrick; brick0 += gridDim.x) {
        int brick = brick0;
        const int bx = brick % nbx; brick /= nbx;
        const int by = brick % nby; brick /= nby;
        const int bz = brick % nbz;
        const int gz0 = bz * IW_TZ - 2, gy0 = by * IC_TY - 2;
        const bool inner_zy = gz0 >= 0 && gz0 + IW_IZ <= D && gy0 >= 0 && gy0 + IC_IY <= H;
        __syncthreads();                             // the previous brick's tiles are read
#pragma unroll
        for (int k = 0; k < IW_NI; ++k) if (tid + 256 * k < IW_TILE) timg[tid + 256 * k] = iv[k];
#pragma unroll
        for (int k = 0; k < IW_ND; ++k) {
            const int q = tid + 256 * k;
            *reinterpret_cast<float4*>(tdy + (size_t)(q >> 2) * 16 + 4 * (q & 3)) = dv[k];
        }
        __syncthreads();
        if (brick0 + (int)gridDim.x < nbrick) {
            ic_issue_brick<O>(dv, iv, img, dy, brick0 + gridDim.x, nbz, nby, nbx, D, H, W, tid);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- image channel: wave = y row of the brick, K steps = 4 consecutive x
#pragma unroll 1
        for (int z = 0; z < IW_TZ; ++z) {
            const float* ta = timg + (z * IC_IY + wave) * IC_IX;
            const float* tbp = tdy + (size_t)((z * IC_TY + wave) * IC_TX + kk) * 16 + i;
#pragma unroll 4
            for (int x4 = 0; x4 < IC_TX / 4; ++x4) {
                const float bv = tbp[x4 * 4 * 16];
                float av[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) av[m] = ta[toff[m] + x4 * 4];
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv, acc[m], 0, 0, 0);
            }
        }
        // ---- indicator channel: T[row][dx][o] = sum_{x: 0 <= x + dx - 2 < W} dy[row][x][o]
        if (tid < NROW * O) {
            const int r = tid / O, o = tid - r * O;
            const float* rowp = tdy + (size_t)(r * IC_TX) * 16 + o;
            // (all 64 reads in flight, four partial sums: a load-add chain costs one LDS round trip per element)
            float sr4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int x = 0; x < IC_TX; ++x) sr4[x & 3] += rowp[(size_t)x * 16];
            const float srow = (sr4[0] + sr4[1]) + (sr4[2] + sr4[3]);
            // edge columns that fall into this brick (absolute x = 0, 1, W - 2, W - 1); voxels beyond W are zero in the tile
            const int x0 = bx * IC_TX;
            auto col = [&](int ox) { const int xl = ox - x0; return (xl >= 0 && xl < IC_TX && ox >= 0 && ox < W) ? rowp[(size_t)xl * 16] : 0.f; };
            const float lo0 = col(0), lo1 = col(1), hi0 = col(W - 1), hi1 = col(W - 2);
            // dx: excluded are x < 2 - dx and x > W + 1 - dx
            tT[(r * 5 + 0) * O + o] = srow - lo0 - (W > 1 ? lo1 : 0.f);
            tT[(r * 5 + 1) * O + o] = srow - lo0;
            tT[(r * 5 + 2) * O + o] = srow;
            tT[(r * 5 + 3) * O + o] = srow - hi0;
            tT[(r * 5 + 4) * O + o] = srow - hi0 - (W > 1 ? hi1 : 0.f);
        }
        __syncthreads();
        if (inner_zy) {
            if (tid < 5 * O) {                       // brick sums per dx
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < NROW; ++r) s += tT[r * 5 * O + tid];
                tT[NROW * 5 * O + tid] = s;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = tid + 256 * j;         // output (tap, o)
                if (q < 125 * O) { const int o = q % O, tap = q / O; g2[j] += tT[(NROW * 5 + tap % 5) * O + o]; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = tid + 256 * j;
                if (q < 125 * O) {
                    const int o = q % O, tap = q / O, dx = tap % 5, dyy = (tap / 5) % 5, dz = tap / 25;
                    float s = 0.f;
                    for (int r = 0; r < NROW; ++r) {
                        const int oz = bz * IW_TZ + r / IC_TY + dz - 2, oy = by * IC_TY + r % IC_TY + dyy - 2;
                        if ((unsigned)oz < (unsigned)D && (unsigned)oy < (unsigned)H) s += tT[(r * 5 + dx) * O + o];
                    }
                    g2[j] += s;
                }
            }
        }
    }
    float* slab = part + (size_t)blockIdx.x * 25 * 16 * O;
    // indicator channel: one owner per output; the unused virtual channels 10..15 are zero
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = tid + 256 * j;
        if (q < 125 * O) { const int o = q % O, tap = q / O; slab[((tap / 5) * 16 + 2 * (tap % 5) + 1) * O + o] = g2[j]; }
    }
    for (int q = tid; q < 25 * 6 * O; q += 256) { const int o = q % O, vch = 10 + (q / O) % 6, t25 = q / (6 * O); slab[(t25 * 16 + vch) * O + o] = 0.f; }
    // image channel: the four waves' partial tiles meet in LDS [wave][tap 128][16 o]; lane (column n = o i, kk) holds rows 4 kk .. 4 kk + 3
    __syncthreads();
    float* redg = smem;
    static_assert(4 * 128 * 16 <= IW_TILE + (NROW + 1) * 5 * 16 + NVB * 16, "reduction buffer fits the tiles");
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) redg[(wave * 128 + 16 * m + 4 * kk + k) * 16 + i] = acc[m][k];
    __syncthreads();
    for (int q = tid; q < 125 * O; q += 256) {
        const int o = q % O, tap = q / O;
        const float s4 = (redg[(0 * 128 + tap) * 16 + o] + redg[(1 * 128 + tap) * 16 + o]) + (redg[(2 * 128 + tap) * 16 + o] + redg[(3 * 128 + tap) * 16 + o]);
        slab[((tap / 5) * 16 + 2 * (tap % 5)) * O + o] = s4;
    }
}

// G[q] = sum over the workgroups' slabs in a FIXED association: four interleaved partial sums (slabs k = u mod 4, in increasing order; the
// leftover slabs of a count that is no multiple of 4 go to partial 0), then (s0 + s1) + (s2 + s3).  A block = 64 outputs x the 4 partial
// sums, one thread each with sixteen loads in flight; the partials meet in LDS.  (One thread per output walking all 512 slabs alone: 41 us
// of load latency; the association is the one of that first version, so the results are bit-identical to it.)
__global__ void __launch_bounds__(256) input_wgrad_reduce_kernel(const float* __restrict__ part, int nslab, int n, float* __restrict__ G) {
    __shared__ float sh[4][64];
    const int ql = threadIdx.x & 63, u = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + ql;
    float s = 0.f;
    if (q < n) {
        const int nmain = nslab & ~3;
        int k = u;
        for (; k + 60 < nmain; k += 64) {
            float t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = part[(size_t)(k + 4 * j) * n + q];
#pragma unroll
            for (int j = 0; j < 16; ++j) s += t[j];
        }
        for (; k < nmain; k += 4) s += part[(size_t)k * n + q];
        if (u == 0) for (k = nmain; k < nslab; ++k) s += part[(size_t)k * n + q];
    }
    sh[u][ql] = s;
    __syncthreads();
    if (u == 0 && q < n) G[q] = (sh[0][ql] + sh[1][ql]) + (sh[2][ql] + sh[3][ql]);
}

}  // namespace

extern "C" {

int vnet_tile_im2col_x(const float* img, float* xv, int B, int D, int H, int W, void* stream) {
    if (!img || !xv || B <= 0 || D <= 0 || H <= 0 || W <= 0) return VNET_E_BADARG;
    const size_t nvox = (size_t)B * D * H * W;
    const size_t nq = nvox * 4;
    const int blocks = (int)(nq / 256 / 4 + 1 > 4096 ? 4096 : nq / 256 / 4 + 1);
    hipLaunchKernelGGL(im2col_x_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, xv, nvox, W);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}


int vnet_input_conv_direct_ok(int O, int B, int D, int H, int W) { return (O == 16 || O == 8) && B > 0 && D > 0 && H > 0 && W > 0 ? 1 : 0; }
int vnet_input_conv_direct_stats_rows(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return B * ceil_div(D, IC_TZ) * ceil_div(H, IC_TY) * ceil_div(W, IC_TX);
}
// border tables of the indicator channel: wbc [9][25][O], cbc [9][O] (x classes (min(x, 2), min(W - 1 - x, 2)); independent of the volume)
int vnet_input_conv_fold_border(const float* wv, int O, float* wbc, float* cbc, void* stream) {
    if (!wv || !wbc || !cbc || O <= 0) return VNET_E_BADARG;
    if (O > 16) return VNET_E_UNSUPPORTED;
    hipLaunchKernelGGL(input_fold_border_kernel, dim3(9), dim3(256), 0, (hipStream_t)stream, wv, O, wbc, cbc);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}
int vnet_input_conv_direct_fwd(const float* img, const float* wv, const float* wbc, const float* cbc, const float* bias, const float* res,
                               float* y, float* stats, int O, int B, int D, int H, int W, void* stream) {
    if (!img || !wv || !wbc || !cbc || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0) return VNET_E_BADARG;
    if (O != 16 && O != 8) return VNET_E_UNSUPPORTED;
    ICArgs a{img, wv, wbc, cbc, bias, res, y, stats, B, D, H, W, ceil_div(D, IC_TZ), ceil_div(H, IC_TY), ceil_div(W, IC_TX)};
    const int grid = B * a.nbz * a.nby * a.nbx;
    if (O == 16) hipLaunchKernelGGL(input_conv_direct_kernel<16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(input_conv_direct_kernel<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}
// workgroups (= partial slabs [25][16][O] floats) of the direct filter gradient; ws must hold that many slabs
static int device_cus_ib() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n;
}
int vnet_input_wgrad_direct_slabs(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const int nbrick = B * ceil_div(D, IW_TZ) * ceil_div(H, IC_TY) * ceil_div(W, IC_TX);
    const int cap = 2 * device_cus_ib();              // two workgroups per CU (49 KB of LDS each; 256 registers per lane)
    return nbrick < cap ? nbrick : cap;
}
int vnet_input_wgrad_direct(const float* img, const float* dy, float* G, int O, int B, int D, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!img || !dy || !G || B <= 0 || D <= 0 || H <= 0 || W <= 0) return VNET_E_BADARG;
    if (O != 16 && O != 8) return VNET_E_UNSUPPORTED;
    const int nbz = ceil_div(D, IW_TZ), nby = ceil_div(H, IC_TY), nbx = ceil_div(W, IC_TX);
    const int grid = vnet_input_wgrad_direct_slabs(B, D, H, W);
    const int n = 25 * 16 * O;
    if (!ws || ws_bytes < (size_t)grid * n * sizeof(float)) return VNET_E_WORKSPACE;
    float* part = reinterpret_cast<float*>(ws);
    const size_t lds = (size_t)(IW_TILE + (IW_TZ * IC_TY + 1) * 5 * 16 + IW_NVB * 16) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (O == 16) {
        static bool done = false;
        if (!done) { if (hipFuncSetAttribute((const void*)input_wgrad_direct_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VNET_E_UNSUPPORTED; done = true; }
        hipLaunchKernelGGL(input_wgrad_direct_kernel<16>, dim3(grid), dim3(256), lds, st, img, dy, part, B, D, H, W, nbz, nby, nbx);
    } else {
        static bool done = false;
        if (!done) { if (hipFuncSetAttribute((const void*)input_wgrad_direct_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return VNET_E_UNSUPPORTED; done = true; }
        hipLaunchKernelGGL(input_wgrad_direct_kernel<8>, dim3(grid), dim3(256), lds, st, img, dy, part, B, D, H, W, nbz, nby, nbx);
    }
    VNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(input_wgrad_reduce_kernel, dim3(ceil_div(n, 64)), dim3(256), 0, st, part, grid, n, G);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

int vnet_input_conv_fold(const float* w, const float* gamma, const float* beta, const float* mean, const float* invstd,
                         float* wv, int C, int O, void* stream) {
    if (!w || !gamma || !beta || !mean || !invstd || !wv || C <= 0 || O <= 0) return VNET_E_BADARG;
    if (O > 16) return VNET_E_UNSUPPORTED;
    hipLaunchKernelGGL(fold_weights_kernel, dim3(ceil_div(25 * 16 * O, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, gamma, beta, mean, invstd, wv, C, O);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

int vnet_input_conv_grads(const float* G, const float* w, const float* gamma, const float* beta, const float* mean,
                          const float* invstd, float* dw, float* dgamma, float* dbeta, int C, int O, int accumulate, void* stream) {
    if (!G || !w || !gamma || !beta || !mean || !invstd || !dw || !dgamma || !dbeta || C <= 0 || O <= 0) return VNET_E_BADARG;
    if (O > 16) return VNET_E_UNSUPPORTED;
    hipLaunchKernelGGL(input_grads_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, G, w, gamma, beta, mean, invstd,
                       dw, dgamma, dbeta, C, O, accumulate);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------
// hard segmentation metrics (reference model.py:588-626): K x K confusion matrix of (label, prediction);
// accuracy / per-class tp, tn, fp, fn / sensitivity / specificity / hard Dice follow on the host.
// ------------------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) confusion_kernel(const long long* __restrict__ pred, const int32_t* __restrict__ labels,
                                                        size_t n, int K, float* __restrict__ partial) {
    __shared__ unsigned int cm[64];
    if (threadIdx.x < 64) cm[threadIdx.x] = 0u;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int l = labels[i], p = (int)pred[i];
        if (l >= 0 && l < K && p >= 0 && p < K) atomicAdd(&cm[l * K + p], 1u);
    }
    __syncthreads();
    if (threadIdx.x < K * K) partial[(size_t)blockIdx.x * K * K + threadIdx.x] = (float)cm[threadIdx.x];
}
__global__ void __launch_bounds__(256) confusion_finalize_kernel(const float* __restrict__ partial, int nblk, int KK, double* __restrict__ out) {
    __shared__ double shd[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += (double)partial[(size_t)b * KK + c];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = shd[0] + shd[1] + shd[2] + shd[3];
}
// tf.metrics.auc (reference model.py:607,613: per class i > 0, labels = one-hot[..., i], predictions = softmax[..., i], default
// num_thresholds = 200, ROC, trapezoidal): TF counts tp/fn/tn/fp at every threshold with `prediction > threshold` (float32).
// Equivalent and one pass: bin(p) = number of thresholds strictly below p (0..T), one histogram for the voxels of the class
// and one for the others; tp[t] = sum of hist_pos[b] over b > t etc. follow on the host.  Integer LDS atomics: exact counts.
__global__ void __launch_bounds__(256) auc_hist_kernel(const float* __restrict__ sm, const int32_t* __restrict__ labels, size_t n, int K, int cls,
                                                       const float* __restrict__ thr, int T, unsigned int* __restrict__ partial) {
    extern __shared__ unsigned int lds_u[];
    unsigned int* hist = lds_u;                       // [2][T + 1]
    float* th = reinterpret_cast<float*>(lds_u + 2 * (T + 1));
    for (int i = threadIdx.x; i < 2 * (T + 1); i += blockDim.x) hist[i] = 0u;
    for (int i = threadIdx.x; i < T; i += blockDim.x) th[i] = thr[i];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float p = sm[i * K + cls];
        int lo = 0, hi = T;                           // smallest b with !(th[b] < p)  ==  count of thresholds below p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (th[mid] < p) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&hist[(labels[i] == cls ? 0 : T + 1) + lo], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (T + 1); i += blockDim.x) partial[(size_t)blockIdx.x * 2 * (T + 1) + i] = hist[i];
}
__global__ void __launch_bounds__(256) auc_hist_finalize_kernel(const unsigned int* __restrict__ partial, int nblk, int W, double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)partial[(size_t)b * W + c];
    out[c] = s;
}
}  // namespace

extern "C" {
size_t vnet_auc_ws_bytes(int T) { return (size_t)256 * 2 * (T + 1) * sizeof(unsigned int); }

int vnet_auc_histogram(const float* softmax, const int32_t* labels, int64_t n, int K, int cls, const float* thresholds, int T,
                       double* hist_out, void* ws, size_t ws_bytes, void* stream) {
    if (!softmax || !labels || !thresholds || !hist_out || n <= 0 || K <= 0 || cls < 0 || cls >= K || T <= 0) return VNET_E_BADARG;
    if (T > 4096) return VNET_E_UNSUPPORTED;
    if (!ws || ws_bytes < vnet_auc_ws_bytes(T)) return VNET_E_WORKSPACE;
    const int64_t want = (n + 256 * 16 - 1) / (256 * 16);
    const int nblk = (int)(want > 256 ? 256 : want);
    const size_t lds = (size_t)(2 * (T + 1) + T) * 4;
    hipLaunchKernelGGL(auc_hist_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, softmax, labels, (size_t)n, K, cls, thresholds, T,
                       (unsigned int*)ws);
    VNET_LAUNCH_CHECK();
    const int W = 2 * (T + 1);
    hipLaunchKernelGGL(auc_hist_finalize_kernel, dim3((W + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned int*)ws, nblk, W, hist_out);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

size_t vnet_confusion_ws_bytes(int K) { return (size_t)1024 * K * K * sizeof(float); }

int vnet_confusion_matrix(const int64_t* pred, const int32_t* labels, int64_t n, int K, double* cm_out,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!pred || !labels || !cm_out || n <= 0 || K <= 0) return VNET_E_BADARG;
    if (K > 8) return VNET_E_UNSUPPORTED;
    if (!ws || ws_bytes < vnet_confusion_ws_bytes(K)) return VNET_E_WORKSPACE;
    const int nblk = (int)((n + 256 * 16 - 1) / (256 * 16) > 1024 ? 1024 : (n + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(confusion_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const long long*)pred, labels, (size_t)n, K, (float*)ws);
    VNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(confusion_finalize_kernel, dim3(K * K), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, K * K, cm_out);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}
}  // extern "C"

// conv_b16.hip -- bf16-STORAGE entry points of the convolution family (BASELINE config C5 as SURVEY 8(d) states it: bf16
// activations and weights into the matrix cores, fp32 accumulation).  Activations, skip tensors and their gradients are bf16
// tensors in HBM (NDHWC, 2 bytes per element); parameter gradients, batch-norm statistics and biases stay fp32.
//   * 5^3 stride-1 convolution (forward and backward-data): the bf16 MFMA kernels of conv_kernels.h with bf16 sources (H) and
//     bf16 outputs (O16): y = RNE(fp32 accumulator + bias [+ stored gradient]) -- bit-for-bit the rounding of what
//     vnet_conv_fwd_bf16_x16 writes in fp32 (same kernels, same summation order);
//   * 5^3 filter gradient: wgrad5_bf16_kernel on the bf16 tensors, fp32 dw;
//   * 2^3 stride-2 down convolution / 2^3 transposed convolution and their filter gradient: the fp32 MFMA kernels with bf16
//     tensors converted while staging and a packed filter of bf16-rounded values (VNET_PACK_ROUND_BF16): every product is the
//     exact bf16 x bf16 product, accumulated in fp32 -- the arithmetic of the matrix cores' bf16 path; these launches are
//     HBM-bound, what matters is that they move 2-byte elements.
// Reference call sites replaced: layers2.py:59-63, 65-74, 78-94 (forward), model.py:660 (their gradients).
#include "conv_kernels.h"
#include "conv_deep.h"
#include "wgrad_zs.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

// workspace bytes / statistics rows of the generic (non-deep) bf16 kernels: functions of the shape only (through round 4 these were
// entry points of the retired fp32-tensor / bf16-operand mode: vnet_conv_bf16_ws_bytes, vnet_conv_bf16_stats_rows_x16)
size_t conv_bf16_generic_ws_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    Bf16Plan p = plan_conv_bf16(Cin, Cout, B, D, H, W);
    if (p.nsplit * p.nz <= 1) return 0;
    return (size_t)p.nsplit * p.nz * B * D * H * W * round_up(Cout, 32) * sizeof(float);
}

int conv_bf16_generic_stats_rows(int Cin, int Cy0, int Cy1, int C0, int C1, int B, int D, int H, int W) {
    if (Cy1 != 0 || Cy0 <= 0 || (Cy0 & 3) || Cin <= 0 || B <= 0) return 0;
    if (conv_bf16_use_c16(Cin, Cy0, C0, C1, Cy0, 0, B, D, H, W)) return B * ceil_div(D, 4) * ceil_div(H, 8) * ceil_div(W, 16);
    Bf16Plan p = plan_conv_bf16(Cin, Cy0, B, D, H, W);
    if (conv_bf16_use_r32(Cy0, Cy0, 0, B, D, H, W) && p.nsplit * p.nz == 1)
        return B * ceil_div(D, 4) * ceil_div(H, 16) * ceil_div(W, 16);      // row-pair kernel: one row per 4x16x16 brick
    if (p.nsplit * p.nz > 1) {
        if (Cy0 > 256 || 256 % Cy0) return 0;
        const size_t total = (size_t)B * D * H * W * Cy0;
        return (int)min((size_t)2048, (total + 255) / 256);
    }
    return B * p.nbz * p.nby * p.nbx;
}
}  // namespace

extern "C" {

// workspace bytes / epilogue-statistics rows of vnet_conv_fwd_b16 for one problem (the kernel choice is a function of the shape)
size_t vnet_conv_b16_ws_bytes(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) {
    // (the larger of the two plans: the zero-padded network input takes the x-im2col kernel whatever the deep plan says)
    const size_t g = conv_bf16_generic_ws_bytes(C0 + C1, Cy0 + Cy1, B, D, H, W);
    const DeepPlan dp = plan_conv_deep(C0, C1, Cy0, Cy1, B, D, H, W, true);      // (whatever VNET_BF16_DEEP says: callers cache this)
    const size_t d = (dp.use && dp.nsplit > 1) ? (size_t)dp.nsplit * B * D * H * W * round_up(Cy0 + Cy1, 32) * sizeof(float) : 0;
    return d > g ? d : g;
}

int vnet_conv_b16_stats_rows(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) {
    if (Cy1 != 0 || Cy0 <= 0 || (Cy0 & 3) || C0 <= 0 || B <= 0) return 0;
    const DeepPlan dp = plan_conv_deep(C0, C1, Cy0, Cy1, B, D, H, W);
    if (!dp.use) return conv_bf16_generic_stats_rows(C0 + C1, Cy0, Cy1, C0, C1, B, D, H, W);
    if (dp.nsplit > 1) {                              // statistics from the split-K reduce kernel: one row per reduce block
        if (Cy0 > 256 || 256 % Cy0) return 0;
        const size_t total = (size_t)B * D * H * W * Cy0;
        return (int)min((size_t)2048, (total + 255) / 256);
    }
    return B * dp.nbz * dp.nby * dp.nbx;              // one row per 4x8x8 brick
}

static int conv_fwd_b16_impl(const void* x0, int C0, const void* x1, int C1, const void* wp, const float* bias,
                             void* y0, int Cy0, void* y1, int Cy1, int B, int D, int H, int W,
                             const void* acc16, const void* res16, float* stats, void* ws, size_t ws_bytes, void* stream, int cin_real);

int vnet_conv_fwd_b16(const void* x0, int C0, const void* x1, int C1, const void* wp, const float* bias,
                      void* y0, int Cy0, void* y1, int Cy1, int B, int D, int H, int W,
                      const void* acc16, const void* res16, float* stats, void* ws, size_t ws_bytes, void* stream) {
    return conv_fwd_b16_impl(x0, C0, x1, C1, wp, bias, y0, Cy0, y1, Cy1, B, D, H, W, acc16, res16, stats, ws, ws_bytes, stream, 0);
}

int vnet_conv_fwd_b16_padded(const void* x16, int Cpad, int Cin, const void* wp, const float* bias, void* y16, int Cout,
                             int B, int D, int H, int W, const void* res16, float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (Cin <= 0 || Cin > Cpad) return VNET_E_BADARG;
    return conv_fwd_b16_impl(x16, Cpad, nullptr, 0, wp, bias, y16, Cout, nullptr, 0, B, D, H, W, nullptr, res16, stats, ws, ws_bytes, stream, Cin);
}

static int conv_fwd_b16_impl(const void* x0, int C0, const void* x1, int C1, const void* wp, const float* bias,
                             void* y0, int Cy0, void* y1, int Cy1, int B, int D, int H, int W,
                             const void* acc16, const void* res16, float* stats, void* ws, size_t ws_bytes, void* stream, int cin_real) {
    if (!x0 || !wp || !y0 || C0 <= 0 || Cy0 <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return VNET_E_BADARG;
    if ((C1 > 0 && !x1) || (Cy1 > 0 && !y1) || C1 < 0 || Cy1 < 0) return VNET_E_BADARG;
    if (stats && Cy1 > 0) return VNET_E_BADARG;
    if ((C0 & 7) || (C1 & 7) || (Cy0 & 3) || (Cy1 & 3) || !al16p(x0) || !al16p(x1) || !al8p(y0) || !al8p(y1) || !al8p(acc16) || !al8p(res16))
        return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a{};
    a.x0 = reinterpret_cast<const float*>(x0); a.x1 = reinterpret_cast<const float*>(x1); a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1;
    a.wp = reinterpret_cast<const float4*>(wp); a.bias = bias;
    a.y0 = reinterpret_cast<float*>(y0); a.y1 = reinterpret_cast<float*>(y1); a.Cy0 = Cy0; a.Cy1 = Cy1; a.Cout = Cy0 + Cy1;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W; a.Do = D; a.Ho = H; a.Wo = W;
    a.nchunks = round_up(a.Cin, 16) / 16; a.CQ = a.nchunks * 4;
    a.CoutP = round_up(a.Cout, 32);
    a.vec_in = 1; a.vec_out = 1;
    a.pad = 2; a.padx = 2; a.accum = acc16 ? 1 : 0; a.res = reinterpret_cast<const float*>(res16); a.stats = stats;
    if (acc16 && acc16 != y0) {
        if (Cy1 > 0) return VNET_E_BADARG;
        a.accsrc = reinterpret_cast<const float*>(acc16);
    }
    if (stats && vnet_conv_b16_stats_rows(C0, C1, Cy0, Cy1, B, D, H, W) == 0) return VNET_E_UNSUPPORTED;
    // the caller vouches that channels cin_real .. C0-1 of x0 are zero (the cast network input): x-im2col form of the 16-cout kernel
    a.in4 = (cin_real > 0 && cin_real <= 4 && C0 == 8 && C1 == 0) ? 1 : 0;
    Bf16Plan p = plan_conv_bf16(a.Cin, a.Cout, B, D, H, W);
    // deep levels (few bricks, wide channels): the K-split-over-waves kernel of conv_deep.h
    const bool al16_all = al16p(y0) && al16p(y1) && al16p(acc16) && al16p(res16);          // (its epilogue moves 16 bytes per lane)
    // (vnet_conv_b16_stats_rows sizes the caller's buffer for the deep kernel's bricks wherever plan_conv_deep takes the
    //  shape; tensors that are only 8-byte aligned would silently fall back to the generic kernels, which write another number of
    //  partial rows -- refuse that combination instead of folding unwritten rows into the batch-norm moments)
    if (stats && !a.in4 && !al16_all && plan_conv_deep(C0, C1, Cy0, Cy1, B, D, H, W).use) return VNET_E_UNSUPPORTED;
    const DeepPlan dp = (a.in4 || !al16_all) ? DeepPlan{} : plan_conv_deep(C0, C1, Cy0, Cy1, B, D, H, W);
    if (dp.use) { a.nbz = dp.nbz; a.nby = dp.nby; a.nbx = dp.nbx; a.cps = dp.cps; a.nz = 1; }
    else { a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.cps = p.cps; a.nz = p.nz; }
    const int nslab = dp.use ? dp.nsplit : p.nsplit * p.nz;
    const size_t nvox = (size_t)B * D * H * W;
    if (nslab > 1) {
        const size_t need = (size_t)nslab * nvox * a.CoutP * sizeof(float);
        if (!ws || ws_bytes < need) return VNET_E_WORKSPACE;
        a.part = reinterpret_cast<float*>(ws); a.part_stride = nvox * a.CoutP;
    }
    const int e = dp.use ? launch_conv_deep(a, dp, st) : conv_fwd_bf16_go<>(a, p, nslab, C0, C1, Cy0, Cy1, B, D, H, W, st);
    if (e == -1) return VNET_OK;
    if (e) return e;
    if (nslab > 1) {
        const size_t total = nvox * a.Cout;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3(blocks), dim3(256), 0, st, a.part, a.part_stride, nslab, bias,
                           reinterpret_cast<unsigned short*>(y0), reinterpret_cast<unsigned short*>(y1), Cy0, Cy1, a.CoutP, nvox, a.accum,
                           reinterpret_cast<const unsigned short*>(res16), a.stats, reinterpret_cast<const unsigned short*>(a.accsrc));
        VNET_LAUNCH_CHECK();
    }
    return VNET_OK;
}

int vnet_conv_wgrad_b16(const void* x0, int C0, const void* x1, int C1, const void* dy, int Cout, float* dw, int Cin_dw,
                        int B, int D, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!x0 || !dy || !dw || C0 <= 0 || Cout <= 0 || B <= 0 || C1 < 0 || (C1 > 0 && !x1)) return VNET_E_BADARG;
    if (D <= 0 || H <= 0 || W <= 0 || Cin_dw <= 0 || Cin_dw > C0 + C1) return VNET_E_BADARG;
    if ((C0 & 7) || (C1 & 7) || (Cout & 7) || !al16p(x0) || !al16p(x1) || !al16p(dy)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a{};
    a.x0 = reinterpret_cast<const float*>(x0); a.x1 = reinterpret_cast<const float*>(x1); a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1;
    a.dy = reinterpret_cast<const float*>(dy); a.Cout = Cout;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W; a.Do = D; a.Ho = H; a.Wo = W;
    a.CinP = round_up(a.Cin, 16); a.CoutP = round_up(Cout, 16);
    a.pad = 2; a.padx = 2; a.vec_in = 1; a.vec_dy = 1;
    // z-streaming kernel (wgrad_zs.h; VNET_WGRAD_ZS=1): 16 cin x 32 cout per workgroup, column steps split over workgroups
    {
        const bool in4z = Cin_dw <= 4 && C0 == 8 && C1 == 0;
        if (tuning().wgrad_zs == 1 && zs_shape_ok(C0, C1, Cout) && zs_depth_ok(D, W) && !in4z && (size_t)D * H * W * max(max(C0, C1), Cout) < ((size_t)1 << 31)) {
            const int nitems = zs_geometry(a);
            const int nblock = (a.CinP / 16) * a.ncob;
            const size_t slab = (size_t)125 * a.CinP * a.CoutP * sizeof(float);
            int ns = max(1, min(nitems, ceil_div(256, nblock)));
            const bool padded = !(a.CinP == Cin_dw && a.CoutP == Cout);
            if (ns > 1 || padded) {
                if (!ws || ws_bytes < slab) return VNET_E_WORKSPACE;
                ns = (int)min((size_t)ns, ws_bytes / slab);
                a.part = reinterpret_cast<float*>(ws);
            } else a.part = dw;
            a.nsplit = ns;
            if (int e = launch_wgrad_zs(a, st)) return e;
            if (a.part != dw) launch_wgrad_reduce(a.part, ns, 125, a.CinP, a.CoutP, Cin_dw, Cout, dw, st);
            VNET_LAUNCH_CHECK();
            return VNET_OK;
        }
    }
    WgradPlan p = plan_wgrad(5, 5, 1, a.Cin, Cout, B, D, H, W, true);
    a.ncob = p.ncob; a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.nbrick = p.nbrick; a.nsplit = p.nsplit;
    const size_t need = (size_t)p.nsplit * 125 * a.CinP * a.CoutP * sizeof(float);
    // one slab and no channel padding: the slab IS dw (TF layout [tap][Cin][Cout]) -> no reduce pass
    bool direct = p.nsplit == 1 && a.CinP == Cin_dw && a.CoutP == Cout;
    if (!direct && (!ws || ws_bytes < need)) return VNET_E_WORKSPACE;
    a.part = direct ? dw : reinterpret_cast<float*>(ws);
    int e;
    // row-reuse kernel (4 x 8 x 32 bricks, one 16-cout block per workgroup): wide volumes with at least two bricks per workgroup
    const int rr_mode = tuning().wgrad_rr;                 // 0: never, 1 (default): where it pays, 2: wherever it applies (tests)
    const int rr_nbrick = B * ceil_div(D, 4) * ceil_div(H, 8) * ceil_div(W, 32);
    const int rr_base = (a.CinP / 16) * (a.CoutP / 16);
    const int rr_nsplit = max(1, min(rr_nbrick, ceil_div(256, rr_base)));
    if (rr_mode && W >= 32 && H >= 8 && rr_nsplit <= p.nsplit && ((C0 & 15) == 0 || C1 == 0) && (rr_mode == 2 || rr_nbrick >= rr_nsplit)) {
        a.ncob = a.CoutP / 16; a.nbz = ceil_div(D, 4); a.nby = ceil_div(H, 8); a.nbx = ceil_div(W, 32);
        a.nbrick = rr_nbrick; a.nsplit = rr_nsplit;
        direct = rr_nsplit == 1 && a.CinP == Cin_dw && a.CoutP == Cout;
        a.part = direct ? dw : reinterpret_cast<float*>(ws);
        // the zero-padded network input (Cin_dw <= 4 real channels of 8): x-im2col form, 10 instead of 25 tap pairs per k-step
        const bool in4 = Cin_dw <= 4 && C0 == 8 && C1 == 0 && tuning().conv_in4 != 0;
        e = in4 ? launch_wgrad_bf16_rr<4, true>(a, rr_nsplit, a.ncob, st) : launch_wgrad_bf16_rr<4>(a, rr_nsplit, a.ncob, st);
        if (e) return e;
        if (direct) return VNET_OK;
        launch_wgrad_reduce(a.part, rr_nsplit, 125, a.CinP, a.CoutP, Cin_dw, Cout, dw, st);
        VNET_LAUNCH_CHECK();
        return VNET_OK;
    }
    if (p.small) {
        e = p.ns == 2 ? launch_wgrad_bf16<4, 8, 8, 2, 8>(a, p.nsplit, p.ncob, p.ntg, st)
                      : launch_wgrad_bf16<4, 8, 8, 1, 16>(a, p.nsplit, p.ncob, p.ntg, st);
    } else {
        e = p.ns == 2 ? launch_wgrad_bf16<4, 4, 16, 2, 8>(a, p.nsplit, p.ncob, p.ntg, st)
                      : launch_wgrad_bf16<4, 4, 16, 1, 16>(a, p.nsplit, p.ncob, p.ntg, st);
    }
    if (e) return e;
    if (direct) return VNET_OK;
    // (Cin_dw < C0 + C1: the leading input channels only -- a network input that was zero-padded to the 16-byte unit)
    launch_wgrad_reduce(a.part, p.nsplit, 125, a.CinP, a.CoutP, Cin_dw, Cout, dw, st);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

}  // extern "C"

// ---- grouped filter gradients (round 4) ---------------------------------------------------------------------------------------
// One layer's filter gradient at the deep levels (32^3 and below) is a launch of 256 workgroups with ONE or TWO bricks each: the
// first tile load is exposed, nothing is prefetched under the MFMAs, and every workgroup writes its whole accumulator block as a
// split-K slab (256 x 128 KB = 32 MB per layer whatever the filter's size: 390 MB per C5 step, written once and read once by the
// reduce).  The layers of a backward pass whose filter gradients nobody reads before the pass ends are therefore collected and
// launched TOGETHER: each layer gets as many workgroups as its share of the work, a workgroup walks 6-16 bricks with the next
// brick's tiles in flight, and the slabs shrink with the split (nsplit 16 -> 2-5 at 32^3, 4 -> 1 at 16^3: a slab-less layer writes
// dw directly and has nothing to reduce).  Same kernels bodies, same per-workgroup summation order over ascending bricks: a
// layer's result depends on its nsplit only (tests compare against the oracle, not against the ungrouped launch, bit for bit).
namespace {
constexpr int WG_MAXJ = 32;          // (a V-Net of 5 levels: 22 5^3 + 8 2^3 convolutions; the table travels in the 4 KB of kernel arguments)
enum { WG_RR = 0, WG_S16 = 1, WG_S8 = 2, WG_ZS32 = 3, WG_ZS16 = 4, WG_ZS8 = 5,
       WG_K2_W2 = 7, WG_K2_W4 = 8, WG_K2_S2 = 9, WG_K2_S4 = 10 };     // 2^3 stride 2: wide / small bricks x 2 / 4 cout blocks
struct WgradGroupJob {                // what the kernel bodies read of WgradArgs, 96 bytes
    const void* x0; const void* x1; const void* dy; float* part;
    int C0, C1, Cout, B, D, H, W, CinP, CoutP, ncob, nbz, nby, nbx, nbrick, nsplit, fam;
};
struct WgradGroup { int n; unsigned blk0[WG_MAXJ + 1]; WgradGroupJob job[WG_MAXJ]; };
static_assert(sizeof(WgradGroup) <= 4096, "the job table is a kernel argument");

__global__ void __launch_bounds__(512) wgrad5_b16_group_kernel(WgradGroup g) {
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < g.n; ++k) if (blockIdx.x >= g.blk0[k]) j = k;
    const WgradGroupJob& q = g.job[j];
    WgradArgs a;
    a.x0 = reinterpret_cast<const float*>(q.x0); a.x1 = reinterpret_cast<const float*>(q.x1); a.dy = reinterpret_cast<const float*>(q.dy);
    a.part = q.part; a.C0 = q.C0; a.C1 = q.C1; a.Cin = q.C0 + q.C1; a.Cout = q.Cout; a.B = q.B;
    a.Di = a.Do = q.D; a.Hi = a.Ho = q.H; a.Wi = a.Wo = q.W; a.CinP = q.CinP; a.CoutP = q.CoutP; a.ncob = q.ncob;
    a.nbz = q.nbz; a.nby = q.nby; a.nbx = q.nbx; a.nbrick = q.nbrick; a.nsplit = q.nsplit;
    a.pad = 2; a.padx = 2; a.vec_in = 1; a.vec_dy = 1;
    if (q.fam >= WG_K2_W2) {             // 2^3 stride 2: x0 is the fine tensor, dy the coarse one
        a.Do = (q.D + 1) >> 1; a.Ho = (q.H + 1) >> 1; a.Wo = (q.W + 1) >> 1; a.pad = 0; a.padx = 0;
    }
    const unsigned local = blockIdx.x - g.blk0[j];
    const int split = (int)(local % (unsigned)a.nsplit);
    const int rest = (int)(local / (unsigned)a.nsplit);
    const int ny = (a.CinP / 16) * a.ncob;
    switch (q.fam) {
        case WG_RR: wgrad5_bf16_rr_body<4, false>(a, split, rest); break;
        case WG_S16: wgrad5_bf16_body<4, 4, 16, 2, 8>(a, split, rest % ny, rest / ny); break;
        case WG_ZS32: wgrad5_b16_zs_body<32>(a, split, rest); break;
        case WG_ZS16: wgrad5_b16_zs_body<16>(a, split, rest); break;
        case WG_ZS8: wgrad5_b16_zs_body<8>(a, split, rest); break;
        case WG_K2_W2: wgrad_body<2, 2, 2, 4, 16, 2, 1, 2, true>(a, split, rest, 0); break;
        case WG_K2_W4: wgrad_body<2, 2, 2, 4, 16, 4, 1, 2, true>(a, split, rest, 0); break;
        case WG_K2_S2: wgrad_body<2, 2, 2, 8, 8, 2, 1, 2, true>(a, split, rest, 0); break;
        case WG_K2_S4: wgrad_body<2, 2, 2, 8, 8, 4, 1, 2, true>(a, split, rest, 0); break;
        default: wgrad5_bf16_body<4, 8, 8, 2, 8>(a, split, rest % ny, rest / ny); break;
    }
}

struct GroupItem { WgradArgs a; int fam, nblock, nbrick, Cin_dw, T3; double unit; float* dw; void* ws; size_t ws_bytes; };

// the plan and the launch shared by the grouped entry points: workgroups per layer by work share, longest first, slabs / direct
// writes, the reduces of the slabs (queued when vnet_wgrad_defer is on)
template <typename K>
int launch_wgrad_group(std::vector<GroupItem>& items, double rounds, K k, size_t lds, unsigned long long& attr_done, void* stream) {
    if (items.empty()) return VNET_OK;
    hipStream_t st = (hipStream_t)stream;
    // work shares: a workgroup should carry total / (CUs x rounds); a layer block of `nbrick` bricks is split accordingly
    double total = 0.0;
    for (const GroupItem& it : items) total += it.unit * it.nblock * it.nbrick;
    const double target = total / ((double)device_cus() * rounds);
    for (GroupItem& it : items) {
        const size_t slab = (size_t)it.T3 * it.a.CinP * it.a.CoutP * sizeof(float);
        int ns = (int)ceil(it.unit * it.nbrick / target - 1e-9);
        ns = max(1, min(ns, it.nbrick));
        const bool padded = !(it.a.CinP == it.Cin_dw && it.a.CoutP == it.a.Cout);
        if (ns > 1 || padded) {
            const size_t cap = it.ws ? it.ws_bytes / slab : 0;
            if (cap < 1) return VNET_E_WORKSPACE;
            ns = (int)min((size_t)ns, cap);
            it.a.part = reinterpret_cast<float*>(it.ws);
        } else it.a.part = it.dw;
        it.a.nsplit = ns;
    }
    std::stable_sort(items.begin(), items.end(), [](const GroupItem& p, const GroupItem& q) {
        return p.unit * ceil_div(p.nbrick, p.a.nsplit) > q.unit * ceil_div(q.nbrick, q.a.nsplit); });
    if (tuning().group_debug) {
        fprintf(stderr, "[wgrad group] %zu layers, total %.0f units, target %.1f units per workgroup\n", items.size(), total, target);
        for (const GroupItem& it : items)
            fprintf(stderr, "  fam %2d  %3d^3 (D %d) %3d->%3d  blocks %3d  bricks %5d  unit %.3f  nsplit %3d  -> %4d workgroups of %.1f units\n",
                    it.fam, it.a.Wi, it.a.Di, it.a.Cin, it.a.Cout, it.nblock, it.nbrick, it.unit, it.a.nsplit, it.nblock * it.a.nsplit,
                    it.unit * ceil_div(it.nbrick, it.a.nsplit));
    }
    if (int ae = ensure_lds(k, lds, attr_done)) return ae;
    for (size_t i0 = 0; i0 < items.size(); i0 += WG_MAXJ) {
        WgradGroup g{};
        g.n = (int)min((size_t)WG_MAXJ, items.size() - i0);
        unsigned blk = 0;
        for (int q = 0; q < g.n; ++q) {
            const GroupItem& it = items[i0 + q];
            const WgradArgs& w = it.a;
            g.job[q] = WgradGroupJob{w.x0, w.x1, w.dy, w.part, w.C0, w.C1, w.Cout, w.B, w.Di, w.Hi, w.Wi, w.CinP, w.CoutP, w.ncob,
                                     w.nbz, w.nby, w.nbx, w.nbrick, w.nsplit, it.fam};
            g.blk0[q] = blk;
            blk += (unsigned)(it.nblock * it.a.nsplit);
        }
        g.blk0[g.n] = blk;
        hipLaunchKernelGGL(k, dim3(blk), dim3(512), lds, st, g);
        VNET_LAUNCH_CHECK();
    }
    for (const GroupItem& it : items)
        if (it.a.part != it.dw) launch_wgrad_reduce(it.a.part, it.a.nsplit, it.T3, it.a.CinP, it.a.CoutP, it.Cin_dw, it.a.Cout, it.dw, st);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

}  // namespace

extern "C" {

size_t vnet_wgrad_job_bytes(void) { return sizeof(vnet_wgrad_job); }

int vnet_conv_wgrad_b16_group(const vnet_wgrad_job* jobs, int n, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return VNET_E_BADARG;
    std::vector<GroupItem> items;
    const double rounds = tuning().group_rounds;
    // WGRAD_ZS: 0 = round-3 kernel bodies only, 1 = the z-streaming kernel wherever it applies, 2 (default) = below 32 voxels
    // per row only: at 32^3 the row-reuse body with many bricks per workgroup measures faster (profiles/r04_wgrad_group.txt)
    const int zs_mode = tuning().wgrad_zs;
    const bool zs_on = zs_mode != 0;
    const int zs_maxw = zs_mode == 2 ? 31 : (1 << 30);
    for (int q = 0; q < n; ++q) {
        const vnet_wgrad_job& J = jobs[q];
        if (!J.x0 || !J.dy || !J.dw || J.C0 <= 0 || J.Cout <= 0 || J.B <= 0 || J.C1 < 0 || (J.C1 > 0 && !J.x1)) return VNET_E_BADARG;
        if (J.D <= 0 || J.H <= 0 || J.W <= 0 || J.Cin_dw <= 0 || J.Cin_dw > J.C0 + J.C1) return VNET_E_BADARG;
        const int Cin = J.C0 + J.C1, CinP = round_up(Cin, 16), CoutP = round_up(J.Cout, 16);
        if (J.ks != 0 && J.ks != 5 && J.ks != 2) return VNET_E_BADARG;
        if (J.ks == 2) {
            if (J.C1 != 0 || J.Cin_dw != J.C0) return VNET_E_BADARG;
            const int Do = (J.D + 1) / 2, Ho = (J.H + 1) / 2, Wo = (J.W + 1) / 2;
            WgradPlan p = plan_wgrad(2, 2, 2, J.C0, J.Cout, J.B, Do, Ho, Wo);
            const bool k2ok = !(J.C0 & 3) && !(J.Cout & 3) && al8p(J.x0) && al8p(J.dy) && (p.ns == 2 || p.ns == 4) && rounds > 0.0;
            if (!k2ok) {
                const int e = vnet_conv2_wgrad_b16(J.x0, J.C0, J.dy, J.Cout, J.dw, J.B, J.D, J.H, J.W, Do, Ho, Wo, J.ws, J.ws_bytes, stream);
                if (e) return e;
                continue;
            }
            GroupItem it{};
            WgradArgs& a = it.a;
            a.x0 = reinterpret_cast<const float*>(J.x0); a.C0 = J.C0; a.Cin = J.C0; a.dy = reinterpret_cast<const float*>(J.dy); a.Cout = J.Cout;
            a.B = J.B; a.Di = J.D; a.Hi = J.H; a.Wi = J.W; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
            a.CinP = CinP; a.CoutP = CoutP; a.vec_in = 1; a.vec_dy = 1;
            a.ncob = p.ncob; a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.nbrick = p.nbrick;
            it.fam = p.small ? (p.ns == 4 ? WG_K2_S4 : WG_K2_S2) : (p.ns == 4 ? WG_K2_W4 : WG_K2_W2);
            it.nblock = (CinP / 16) * p.ncob; it.nbrick = p.nbrick; it.Cin_dw = J.C0; it.T3 = 8;
            // a brick = 128 coarse voxels: 32 KB of the fine tensor + 4-16 KB of dy, next to no MFMA work to hide the loads behind: the
            // stand-alone launches measure ~4.3 us per brick (latency-bound), a third of a row-reuse brick -- an underestimate leaves
            // a layer to a handful of workgroups (0.02 x ns here cost +1.8 ms of tail in the C5 step)
            it.unit = 0.35;
            it.dw = J.dw; it.ws = J.ws; it.ws_bytes = J.ws_bytes;
            items.push_back(it);
            continue;
        }
        const bool in4 = J.Cin_dw <= 4 && J.C0 == 8 && J.C1 == 0;
        const bool ok = !(J.C0 & 15) && !(J.C1 & 15) && !(J.Cout & 7) && al16p(J.x0) && al16p(J.x1) && al16p(J.dy) && !in4;
        int fam = -1;
        // (the zero-padded network input keeps its own launch -- the x-im2col form of the row-reuse body: inside the group it measured
        //  +0.07 ms on the C5 step, round 4)
        if (in4) fam = -1;
        else if (ok && zs_on && J.W <= zs_maxw && zs_shape_ok(J.C0, J.C1, J.Cout) && zs_depth_ok(J.D, J.W) && (size_t)J.D * J.H * J.W * max(max(J.C0, J.C1), J.Cout) < ((size_t)1 << 31))
            fam = J.W >= 32 ? WG_ZS32 : (J.W >= 16 ? WG_ZS16 : WG_ZS8);
        else if (ok && J.W >= 32 && J.H >= 8) fam = WG_RR;
        else if (ok && (CoutP % 32) == 0) fam = J.W >= 16 ? WG_S16 : WG_S8;
        if (fam < 0 || rounds <= 0.0) {           // not a shape of the grouped kernels: the layer's own launch
            const int e = vnet_conv_wgrad_b16(J.x0, J.C0, J.x1, J.C1, J.dy, J.Cout, J.dw, J.Cin_dw, J.B, J.D, J.H, J.W, J.ws, J.ws_bytes, stream);
            if (e) return e;
            continue;
        }
        GroupItem it{};
        WgradArgs& a = it.a;
        a.x0 = reinterpret_cast<const float*>(J.x0); a.x1 = reinterpret_cast<const float*>(J.x1); a.C0 = J.C0; a.C1 = J.C1; a.Cin = Cin;
        a.dy = reinterpret_cast<const float*>(J.dy); a.Cout = J.Cout;
        a.B = J.B; a.Di = J.D; a.Hi = J.H; a.Wi = J.W; a.Do = J.D; a.Ho = J.H; a.Wo = J.W;
        a.CinP = CinP; a.CoutP = CoutP; a.pad = 2; a.padx = 2; a.vec_in = 1; a.vec_dy = 1;
        const bool zs_fam = fam >= WG_ZS32 && fam <= WG_ZS8;
        if (zs_fam) {
            a.Do = J.D; a.Ho = J.H; a.Wo = J.W;
            const int nitems = zs_geometry(a);
            (void)nitems;
            it.nblock = (CinP / 16) * a.ncob; it.unit = 0.4;                 // a column step: 256 voxels x 16 cin x 32 cout x 125 taps
        } else if (fam == WG_RR) {
            a.ncob = CoutP / 16; a.nbz = ceil_div(J.D, 4); a.nby = ceil_div(J.H, 8); a.nbx = ceil_div(J.W, 32);
            it.nblock = (CinP / 16) * a.ncob; it.unit = 1.0;
        } else {
            a.ncob = CoutP / 32; a.nbz = ceil_div(J.D, 4);
            if (fam == WG_S16) { a.nby = ceil_div(J.H, 4); a.nbx = ceil_div(J.W, 16); }
            else { a.nby = ceil_div(J.H, 8); a.nbx = ceil_div(J.W, 8); }
            it.nblock = (CinP / 16) * a.ncob * 2; it.unit = 0.5;          // two tap groups of 64; a brick is 256 voxels and tile-bound
        }
        if (!zs_fam) a.nbrick = J.B * a.nbz * a.nby * a.nbx;
        it.fam = fam; it.nbrick = a.nbrick; it.Cin_dw = J.Cin_dw; it.T3 = 125; it.dw = J.dw; it.ws = J.ws; it.ws_bytes = J.ws_bytes;
        items.push_back(it);
    }
    static unsigned long long attr_done = 0;
    constexpr size_t LDS_RR = (size_t)8 * 12 * 36 * 32 + (size_t)4 * 8 * 32 * 32;
    return launch_wgrad_group(items, rounds, wgrad5_b16_group_kernel, LDS_RR, attr_done, stream);
}

// 2^3 stride-2 convolution (up = 0: [B,Di,Hi,Wi,Cin] -> [B,Do,Ho,Wo,Cout]) or 2^3 transposed convolution (up = 1) on bf16
// tensors.  wp: vnet_pack_weights(VNET_PACK_FWD | VNET_PACK_ROUND_BF16, 8, Cin, Cout) resp. (VNET_PACK_UP | VNET_PACK_ROUND_BF16).
int vnet_conv2_fwd_b16(int up, const void* x, int Cin, const float* wp, const float* bias, void* y, int Cout,
                       int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                       int accum, float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !wp || !y || Cin <= 0 || Cout <= 0 || B <= 0) return VNET_E_BADARG;
    if (Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return VNET_E_BADARG;
    if ((Cin & 3) || (Cout & 3) || !al8p(x) || !al8p(y)) return VNET_E_UNSUPPORTED;
    if (stats && (up || accum)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a{};
    a.x0 = reinterpret_cast<const float*>(x); a.x1 = nullptr; a.C0 = Cin; a.C1 = 0; a.Cin = Cin;
    a.wp = reinterpret_cast<const float4*>(wp); a.bias = bias;
    a.y0 = reinterpret_cast<float*>(y); a.y1 = nullptr; a.Cy0 = Cout; a.Cy1 = 0; a.Cout = Cout;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.CQ = round_up(Cin, 16) / 4; a.nchunks = a.CQ / 4;
    a.vec_in = 1; a.vec_out = 1; a.accum = accum ? 1 : 0; a.stats = stats; a.pad = 0; a.padx = 0;
    if (stats && vnet_conv_stats_rows(2, 0, 2, 0, Cin, Cout, 0, B, Do, Ho, Wo) == 0) return VNET_E_UNSUPPORTED;
    if (up) { a.CoutP = round_up(8 * Cout, 16); a.upO = Cout; }
    else a.CoutP = round_up(Cout, 16);
    const int gD = up ? Di : Do, gH = up ? Hi : Ho, gW = up ? Wi : Wo;
    ConvPlan p = plan_conv(2, 2, up, Cin, Cout, B, gD, gH, gW, gW);
    a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.cps = p.cps; a.nz = p.nz;
    const int nslab = p.nsplit * p.nz;
    const size_t nvox = (size_t)B * Do * Ho * Wo;
    if (nslab > 1) {
        const size_t need = (size_t)nslab * nvox * a.CoutP * sizeof(float);
        if (!ws || ws_bytes < need) return VNET_E_WORKSPACE;
        a.part = reinterpret_cast<float*>(ws); a.part_stride = nvox * a.CoutP;
    }
    int e;
    if (up) {
        e = p.tiny ? launch_conv_ns<1, 1, 2, 8, 8, 4, 2, true, 1, false, true>(a, p, st)
          : p.small ? launch_conv_ns<1, 1, 8, 8, 8, 8, 4, true, 1, false, true>(a, p, st)
                    : launch_conv_ns<1, 1, 2, 4, 16, 4, 2, true, 1, false, true>(a, p, st);
    } else if (stats && nslab == 1) {
        e = p.small ? launch_conv_ns<2, 2, 2, 8, 8, 4, 2, false, 2, true, true>(a, p, st)
                    : launch_conv_ns<2, 2, 1, 4, 16, 4, 1, false, 2, true, true>(a, p, st);
    } else {
        e = p.small ? launch_conv_ns<2, 2, 2, 8, 8, 4, 2, false, 2, false, true>(a, p, st)
                    : launch_conv_ns<2, 2, 1, 4, 16, 4, 1, false, 2, false, true>(a, p, st);
    }
    if (e) return e;
    if (nslab > 1) {
        const size_t total = nvox * a.Cout;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3(blocks), dim3(256), 0, st, a.part, a.part_stride, nslab, bias,
                           reinterpret_cast<unsigned short*>(y), (unsigned short*)nullptr, Cout, 0, a.CoutP, nvox, a.accum,
                           (const unsigned short*)nullptr, a.stats, (const unsigned short*)nullptr);
        VNET_LAUNCH_CHECK();
    }
    return VNET_OK;
}

// filter gradient of the 2^3 stride-2 convolution on bf16 tensors: x = the FINE tensor [B,Di,Hi,Wi,Cin], dy = the COARSE one
// [B,Do,Ho,Wo,Cout]; dw fp32 [8][Cin][Cout].  (For the transposed convolution call it with the roles it has there: x = its output
// gradient, dy = its input, see vnet_conv_wgrad.)
int vnet_conv2_wgrad_b16(const void* x, int Cin, const void* dy, int Cout, float* dw,
                         int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw || Cin <= 0 || Cout <= 0 || B <= 0) return VNET_E_BADARG;
    if (Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return VNET_E_BADARG;
    if ((Cin & 3) || (Cout & 3) || !al8p(x) || !al8p(dy)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a{};
    a.x0 = reinterpret_cast<const float*>(x); a.x1 = nullptr; a.C0 = Cin; a.C1 = 0; a.Cin = Cin;
    a.dy = reinterpret_cast<const float*>(dy); a.Cout = Cout;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.CinP = round_up(Cin, 16); a.CoutP = round_up(Cout, 16);
    a.pad = 0; a.padx = 0; a.vec_in = 1; a.vec_dy = 1;
    WgradPlan p = plan_wgrad(2, 2, 2, Cin, Cout, B, Do, Ho, Wo);
    a.ncob = p.ncob; a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.nbrick = p.nbrick; a.nsplit = p.nsplit;
    const size_t need = (size_t)p.nsplit * 8 * a.CinP * a.CoutP * sizeof(float);
    const bool direct = p.nsplit == 1 && a.CinP == Cin && a.CoutP == Cout;
    if (!direct && (!ws || ws_bytes < need)) return VNET_E_WORKSPACE;
    a.part = direct ? dw : reinterpret_cast<float*>(ws);
    int e;
    if (p.small) {
        e = p.ns == 4 ? launch_wgrad<2, 2, 2, 8, 8, 4, 1, 2, true>(a, p, st) : p.ns == 2 ? launch_wgrad<2, 2, 2, 8, 8, 2, 1, 2, true>(a, p, st)
                                                                                       : launch_wgrad<2, 2, 2, 8, 8, 1, 1, 2, true>(a, p, st);
    } else {
        e = p.ns == 4 ? launch_wgrad<2, 2, 2, 4, 16, 4, 1, 2, true>(a, p, st) : p.ns == 2 ? launch_wgrad<2, 2, 2, 4, 16, 2, 1, 2, true>(a, p, st)
                                                                                        : launch_wgrad<2, 2, 2, 4, 16, 1, 1, 2, true>(a, p, st);
    }
    if (e) return e;
    if (direct) return VNET_OK;
    launch_wgrad_reduce(a.part, p.nsplit, 8, a.CinP, a.CoutP, Cin, Cout, dw, st);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

}  // extern "C"

#ifdef VNET_STAMPS
// experiment build only: where the persistent bf16 kernels write their s_memtime stamps (4 steps x 8 waves x 12 int64)
extern "C" int vnet_debug_set_stamps(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf));
}
#endif

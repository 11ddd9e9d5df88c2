// conv_mfma.hip -- NDHWC fp32 3-D convolution family on CDNA4 matrix cores (gfx950).
//
// Replaces tf.nn.convolution / tf.nn.conv3d_transpose (reference layers2.py:63,73) and the
// Conv3DBackpropInput/Filter ops TF autodiff builds at model.py:660.  One implicit-GEMM kernel
// serves forward, backward-data (flipped/transposed packed weights) and the 2x2x2 stride-2
// down / transposed-up pair; a second kernel computes filter gradients.
//
// Mapping (exact fp32: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD = the fp32 peak of the chip):
//   D[row = cout][col = voxel] += A[cout][k] * B[k][voxel],   k = (tap, cin)
//   * a workgroup owns a TZxTYxTX brick of output voxels and NS*16 output channels;
//   * the input brick + halo for one 16-channel chunk is staged in LDS once and re-used by all
//     KS^3 taps (zero-filled outside the volume = TF 'SAME');
//   * B fragments are 16-byte LDS reads (4 consecutive cin of one voxel), A fragments are 16-byte
//     global/L2 reads of weights pre-packed as [tap][cin/4][cout][cin%4]; MFMA step j of a
//     k-group uses element j of both, i.e. hardware k-index kk <-> cin 4*kk+j (a K permutation
//     shared by A and B);
//   * each lane ends up with 4 consecutive cout of one voxel -> 16-byte NDHWC stores.
// The same file holds the bf16-operand variants of the 5^3 convolution and of its filter gradient (BASELINE
// config C5: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 with LDS transpose reads) further down.
#include "conv_kernels.h"
#include <string.h>

namespace vnet_detail {
DeferState& defer_state() { static DeferState s; return s; }
static Tuning tuning_from_env() {
    auto geti = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    Tuning t{};
    t.wgrad_zs = geti("VNET_WGRAD_ZS", 2); t.wgrad_rr = geti("VNET_WGRAD_RR", 1); t.conv_in4 = geti("VNET_CONV_IN4", 1);
    const char* g = getenv("VNET_WGRAD_GROUP_ROUNDS");
    t.group_rounds = g ? atof(g) : 2.0;
    t.group_debug = getenv("VNET_WGRAD_GROUP_DEBUG") ? 1 : 0;
    t.bf16_deep = geti("VNET_BF16_DEEP", 1); t.bf16_deep_target = geti("VNET_BF16_DEEP_TARGET", 256);
    t.f32_small = geti("VNET_F32_SMALL", 2); t.x3_nb2 = geti("VNET_X3_NB2", 1); t.bf16_c16pp = geti("VNET_BF16_C16PP", 1);
    return t;
}
Tuning& tuning() { static Tuning t = tuning_from_env(); return t; }
}  // namespace vnet_detail

namespace {
double* option_slot(const char* name, int** ip) {
    vnet_detail::Tuning& t = vnet_detail::tuning();
    *ip = nullptr;
    if (!name) return nullptr;
    if (!strcmp(name, "WGRAD_ZS")) *ip = &t.wgrad_zs;
    else if (!strcmp(name, "WGRAD_RR")) *ip = &t.wgrad_rr;
    else if (!strcmp(name, "CONV_IN4")) *ip = &t.conv_in4;
    else if (!strcmp(name, "WGRAD_GROUP_DEBUG")) *ip = &t.group_debug;
    else if (!strcmp(name, "BF16_DEEP")) *ip = &t.bf16_deep;
    else if (!strcmp(name, "BF16_DEEP_TARGET")) *ip = &t.bf16_deep_target;
    else if (!strcmp(name, "F32_SMALL")) *ip = &t.f32_small;
    else if (!strcmp(name, "X3_NB2")) *ip = &t.x3_nb2;
    else if (!strcmp(name, "BF16_C16PP")) *ip = &t.bf16_c16pp;
    else if (!strcmp(name, "WGRAD_GROUP_ROUNDS")) return &t.group_rounds;
    return nullptr;
}
}  // namespace

extern "C" {

// tuning switches (names = the environment variables without the VNET_ prefix); returns the previous value, NaN for an unknown name
double vnet_set_option(const char* name, double value) {
    int* ip; double* dp = option_slot(name, &ip);
    if (ip) { const double prev = *ip; *ip = (int)value; return prev; }
    if (dp) { const double prev = *dp; *dp = value; return prev; }
    return __builtin_nan("");
}
double vnet_get_option(const char* name) {
    int* ip; double* dp = option_slot(name, &ip);
    return ip ? (double)*ip : (dp ? *dp : __builtin_nan(""));
}

const char* vnet_version(void) { return "vnet_hip 0.2 (gfx950; fp32 MFMA 16x16x4, bf16 MFMA 32x32x16 / 16x16x32)"; }

size_t vnet_packed_weight_floats(int mode, int taps, int I, int O) {
    if (mode == VNET_PACK_FWD_X3 || mode == VNET_PACK_BWD_X3) {         // three bf16 images, size still quoted in floats
        int nchunk, ncob; x3_packed_dims(mode == VNET_PACK_BWD_X3, I, O, &nchunk, &ncob);
        return (size_t)nchunk * X3_NPAIR * ncob * 3 * 256;
    }
    if (mode == VNET_PACK_FWD_BF16 || mode == VNET_PACK_BWD_BF16) {     // bf16 image, size still quoted in floats
        int nchunk, ncob; packed_dims_bf16(mode, I, O, &nchunk, &ncob);
        return (size_t)nchunk * taps * ncob * 512 / 2;
    }
    int Tp, CQ, NP; packed_dims(mode & ~VNET_PACK_ROUND_BF16, taps, I, O, &Tp, &CQ, &NP);
    return (size_t)Tp * CQ * NP * 4;
}

int vnet_pack_weights(int mode, const float* w, float* wp, int taps, int I, int O, void* stream) {
    if (!w || !wp || taps <= 0 || I <= 0 || O <= 0) return VNET_E_BADARG;
    if (mode == VNET_PACK_FWD_X3 || mode == VNET_PACK_BWD_X3) {
        if (taps != 125) return VNET_E_UNSUPPORTED;
        int nchunk, ncob; x3_packed_dims(mode == VNET_PACK_BWD_X3, I, O, &nchunk, &ncob);
        const uint32_t units = (uint32_t)nchunk * X3_NPAIR * ncob * 64;
        hipLaunchKernelGGL(x3_pack_kernel, dim3(min(4096u, (units + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           mode == VNET_PACK_BWD_X3 ? 1 : 0, w, reinterpret_cast<u32x4*>(wp), I, O, ncob, units);
        VNET_LAUNCH_CHECK();
        return VNET_OK;
    }
    if (mode == VNET_PACK_FWD_BF16 || mode == VNET_PACK_BWD_BF16) {
        int nchunk, ncob; packed_dims_bf16(mode, I, O, &nchunk, &ncob);
        const size_t total = (size_t)nchunk * taps * ncob * 512;
        const int blocks = (int)min((size_t)4096, (total + 255) / 256);
        hipLaunchKernelGGL(pack_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mode, w,
                           reinterpret_cast<unsigned short*>(wp), taps, I, O, ncob, total);
        VNET_LAUNCH_CHECK();
        return VNET_OK;
    }
    const int base = mode & ~VNET_PACK_ROUND_BF16;
    if (base < 0 || base > 2) return VNET_E_UNSUPPORTED;
    if (base == VNET_PACK_UP && taps != 8) return VNET_E_UNSUPPORTED;
    int Tp, CQ, NP; packed_dims(base, taps, I, O, &Tp, &CQ, &NP);
    const size_t total = (size_t)Tp * CQ * NP * 4;
    const int blocks = (int)min((size_t)4096, (total + 255) / 256);
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mode, w, wp, taps, I, O, CQ, NP, total);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

int vnet_packed_dims(int mode, int taps, int I, int O, int* CQ, int* NP) {
    if (!CQ || !NP) return VNET_E_BADARG;
    if (mode == VNET_PACK_FWD_X3 || mode == VNET_PACK_BWD_X3) { x3_packed_dims(mode == VNET_PACK_BWD_X3, I, O, CQ, NP); return VNET_OK; }   // (k chunks, n blocks of 16)
    if (mode == VNET_PACK_FWD_BF16 || mode == VNET_PACK_BWD_BF16) { packed_dims_bf16(mode, I, O, CQ, NP); return VNET_OK; }   // (chunks, cout blocks)
    mode &= ~VNET_PACK_ROUND_BF16;
    if (mode < 0 || mode > 2) return VNET_E_BADARG;
    int Tp; packed_dims(mode, taps, I, O, &Tp, CQ, NP);
    return VNET_OK;
}

int vnet_pack_weights_batched(const void* descs_dev, int n, void* stream) {
    if (!descs_dev || n <= 0) return VNET_E_BADARG;
    hipLaunchKernelGGL(pack_batched_kernel, dim3(512, n), dim3(256), 0, (hipStream_t)stream, (const long long*)descs_dev);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

size_t vnet_conv_ws_bytes(int ks, int kx, int stride, int up, int Cin, int Cout, int B, int Do, int Ho, int Wo) {
    (void)kx;
    const int gridW = up ? (Wo + 1) / 2 : Wo;
    ConvPlan p = plan_conv(ks, stride, up, Cin, Cout, B, up ? (Do + 1) / 2 : Do, up ? (Ho + 1) / 2 : Ho, gridW, gridW);
    if (p.nsplit * p.nz <= 1) return 0;
    return (size_t)p.nsplit * p.nz * B * Do * Ho * Wo * round_up(Cout, 16) * sizeof(float);
}

// rows of the epilogue-statistics partial buffer [rows][2][Cout] (0 = this launch cannot produce them: transposed conv, split
// output, channel count that is no multiple of 4, or a split-K reduce whose channel count does not divide 256)
int vnet_conv_stats_rows(int ks, int kx, int stride, int up, int Cin, int Cy0, int Cy1, int B, int Do, int Ho, int Wo) {
    if (up || Cy1 != 0 || Cy0 <= 0 || (Cy0 & 3) || Cin <= 0 || B <= 0) return 0;
    const bool is5 = (ks == 5 && stride == 1), isdown = (ks == 2 && stride == 2);
    if (!is5 && !isdown) return 0;
    if (kx == 0) kx = ks;
    ConvPlan p = plan_conv(ks, stride, 0, Cin, Cy0, B, Do, Ho, Wo, Wo);
    if (p.nsplit * p.nz > 1) {
        if (Cy0 > 256 || 256 % Cy0) return 0;
        const size_t total = (size_t)B * Do * Ho * Wo * Cy0;
        return (int)min((size_t)2048, (total + 255) / 256);
    }
    return B * p.nbz * p.nby * p.nbx;
}

// 1: the statistics of this launch come from the split-K reduce kernel (an HBM-bound pass that already touches every output
// element: they are free there); 0: from the MFMA kernel's own epilogue (the STATS instantiation)
int vnet_conv_stats_from_reduce(int ks, int kx, int stride, int Cin, int Cout, int B, int Do, int Ho, int Wo) {
    const bool is5 = (ks == 5 && stride == 1), isdown = (ks == 2 && stride == 2);
    if ((!is5 && !isdown) || Cin <= 0 || Cout <= 0 || B <= 0) return 0;
    ConvPlan p = plan_conv(ks, stride, 0, Cin, Cout, B, Do, Ho, Wo, Wo);
    return p.nsplit * p.nz > 1 ? 1 : 0;
}

static int conv_fwd_impl(int ks, int kx, int stride, int up, const float* x0, int C0, const float* x1, int C1,
                         const float* wp, const float* bias, float* y0, int Cy0, float* y1, int Cy1,
                         int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                         void* ws, size_t ws_bytes, void* stream, int accum, const float* res = nullptr, float* stats = nullptr) {
    if (!x0 || !wp || !y0 || C0 <= 0 || Cy0 <= 0 || B <= 0) return VNET_E_BADARG;
    if ((C1 > 0 && !x1) || (Cy1 > 0 && !y1) || C1 < 0 || Cy1 < 0) return VNET_E_BADARG;
    if (Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return VNET_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    ConvArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1;
    a.wp = reinterpret_cast<const float4*>(wp); a.bias = bias;
    a.y0 = y0; a.y1 = y1; a.Cy0 = Cy0; a.Cy1 = Cy1; a.Cout = Cy0 + Cy1;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.CQ = round_up(a.Cin, 16) / 4;
    a.nchunks = a.CQ / 4;
    a.vec_in = (C0 % 4 == 0) && (C1 % 4 == 0);
    a.vec_out = (Cy0 % 4 == 0) && (Cy1 % 4 == 0);
    a.part = nullptr; a.part_stride = 0; a.upO = 0; a.accum = accum; a.res = res; a.stats = stats;
    if (stats && vnet_conv_stats_rows(ks, kx, stride, up, C0 + C1, Cy0, Cy1, B, Do, Ho, Wo) == 0) return VNET_E_UNSUPPORTED;
    const bool is5 = (ks == 5 && stride == 1 && !up), isdown = (ks == 2 && stride == 2 && !up), isup = (ks == 2 && stride == 2 && up);
    if (!is5 && !isdown && !isup) return VNET_E_UNSUPPORTED;
    if (kx == 0) kx = ks;
    if (kx != ks && !(is5 && kx == 1 && round_up(Cy0 + Cy1, 16) == 16)) return VNET_E_UNSUPPORTED;   // 5x5x1: x-im2col'ed input conv
    if (isup) {
        if (Cy1 != 0) return VNET_E_UNSUPPORTED;
        a.CoutP = round_up(8 * Cy0, 16); a.upO = Cy0; a.pad = 0; a.padx = 0;
    } else {
        a.CoutP = round_up(a.Cout, 16); a.pad = is5 ? 2 : 0;
        a.padx = (kx - 1) / 2;
    }
    const int gD = isup ? Di : Do, gH = isup ? Hi : Ho, gW = isup ? Wi : Wo;
    ConvPlan p = plan_conv(ks, stride, up, a.Cin, isup ? Cy0 : a.Cout, B, gD, gH, gW, gW);
    a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.cps = p.cps; a.nz = p.nz;
    const int nslab = p.nsplit * p.nz;
    const size_t nvox = (size_t)B * Do * Ho * Wo;
    if (nslab > 1) {
        const size_t need = (size_t)nslab * nvox * a.CoutP * sizeof(float);
        if (!ws || ws_bytes < need) return VNET_E_WORKSPACE;
        a.part = reinterpret_cast<float*>(ws); a.part_stride = nvox * a.CoutP;
    }
    int e;
    if (a.stats && nslab == 1) {     // statistics in the conv epilogue: the STATS instantiations (split-K launches take theirs from the reduce)
        if (is5) {
            if (kx == 1) e = (p.small && !p.half) ? launch_conv_ns<5, 1, 8, 8, 8, 8, 4, false, 1, true>(a, p, st) : launch_conv_ns<5, 1, 4, 8, 8, 4, 4, false, 1, true>(a, p, st);
            else e = p.half == 2 ? launch_conv_ns<5, 1, 4, 4, 4, 4, 1, false, 5, true>(a, p, st)
                   : (p.small && !p.half) ? launch_conv_ns<5, 1, 8, 8, 8, 8, 4, false, 5, true>(a, p, st) : launch_conv_ns<5, 1, 4, 8, 8, 4, 4, false, 5, true>(a, p, st);
        } else {
            e = p.small ? launch_conv_ns<2, 2, 2, 8, 8, 4, 2, false, 2, true>(a, p, st) : launch_conv_ns<2, 2, 1, 4, 16, 4, 1, false, 2, true>(a, p, st);
        }
    } else if (is5) {
        if (kx == 1) e = (p.small && !p.half) ? launch_conv_ns<5, 1, 8, 8, 8, 8, 4, false, 1>(a, p, st) : launch_conv_ns<5, 1, 4, 8, 8, 4, 4, false, 1>(a, p, st);
        else e = p.half == 2 ? launch_conv_ns<5, 1, 4, 4, 4, 4, 1, false>(a, p, st)
               : (p.small && !p.half) ? launch_conv_ns<5, 1, 8, 8, 8, 8, 4, false>(a, p, st) : launch_conv_ns<5, 1, 4, 8, 8, 4, 4, false>(a, p, st);
    } else if (isdown) {
        e = p.small ? launch_conv_ns<2, 2, 2, 8, 8, 4, 2, false>(a, p, st) : launch_conv_ns<2, 2, 1, 4, 16, 4, 1, false>(a, p, st);
    } else {
        e = p.tiny ? launch_conv_ns<1, 1, 2, 8, 8, 4, 2, true>(a, p, st)
          : p.small ? launch_conv_ns<1, 1, 8, 8, 8, 8, 4, true>(a, p, st) : launch_conv_ns<1, 1, 2, 4, 16, 4, 2, true>(a, p, st);
    }
    if (e) return e;
    if (nslab > 1) {
        const size_t total = nvox * a.Cout;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.part, a.part_stride, nslab, bias,
                           y0, y1, Cy0, Cy1, a.CoutP, nvox, a.accum, a.res, a.stats);
        VNET_LAUNCH_CHECK();
    }
    return VNET_OK;
}


int vnet_conv_fwd(int ks, int kx, int stride, int up, const float* x0, int C0, const float* x1, int C1,
                  const float* wp, const float* bias, float* y0, int Cy0, float* y1, int Cy1,
                  int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                  void* ws, size_t ws_bytes, void* stream) {
    return conv_fwd_impl(ks, kx, stride, up, x0, C0, x1, C1, wp, bias, y0, Cy0, y1, Cy1, B, Di, Hi, Wi, Do, Ho, Wo, ws, ws_bytes, stream, 0);
}
int vnet_conv_fwd_acc(int ks, int kx, int stride, int up, const float* x0, int C0, const float* x1, int C1,
                      const float* wp, const float* bias, float* y0, int Cy0, float* y1, int Cy1,
                      int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                      void* ws, size_t ws_bytes, void* stream) {
    return conv_fwd_impl(ks, kx, stride, up, x0, C0, x1, C1, wp, bias, y0, Cy0, y1, Cy1, B, Di, Hi, Wi, Do, Ho, Wo, ws, ws_bytes, stream, 1);
}

int vnet_conv_fwd_stats(int ks, int kx, int stride, const float* x0, int C0, const float* x1, int C1,
                        const float* wp, const float* bias, float* y, int Cout,
                        int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                        const float* res, float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (!stats) return VNET_E_BADARG;
    return conv_fwd_impl(ks, kx, stride, 0, x0, C0, x1, C1, wp, bias, y, Cout, nullptr, 0, B, Di, Hi, Wi, Do, Ho, Wo, ws, ws_bytes, stream, 0,
                         res, stats);
}

}  // extern "C"

extern "C" {

int vnet_wgrad_defer(int on, void* stream) {
    DeferState& ds = defer_state();
    std::lock_guard<std::mutex> lk(ds.mu);
    DeferQueue& q = ds.q[(hipStream_t)stream];
    const int prev = q.on ? 1 : 0;
    q.on = on != 0;
    return prev;
}

int vnet_wgrad_pending(void* stream) {
    DeferState& ds = defer_state();
    std::lock_guard<std::mutex> lk(ds.mu);
    auto it = ds.q.find((hipStream_t)stream);
    return it == ds.q.end() ? 0 : (int)it->second.pending.size();
}

int vnet_wgrad_flush(void* stream) {
    DeferState& ds = defer_state();
    std::vector<ReduceJob> jobs;
    {
        std::lock_guard<std::mutex> lk(ds.mu);
        auto it = ds.q.find((hipStream_t)stream);
        if (it != ds.q.end()) {
            jobs.swap(it->second.pending);
            if (!it->second.on) ds.q.erase(it);              // (streams come and go: no entry outlives its deferring pass)
        }
    }
    hipStream_t st = (hipStream_t)stream;
    for (size_t i0 = 0; i0 < jobs.size(); i0 += REDUCE_BATCH) {
        ReduceBatch b{};
        b.n = (int)min((size_t)REDUCE_BATCH, jobs.size() - i0);
        unsigned blk = 0;
        for (int k = 0; k < b.n; ++k) {
            b.job[k] = jobs[i0 + k];
            b.job[k].blk0 = blk;
            const size_t total = (size_t)b.job[k].T3 * b.job[k].Cin * b.job[k].Cout / (b.job[k].vec ? 4 : 1);
            blk += (unsigned)min((size_t)1024, (total + 63) / 64);
        }
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blk), dim3(256), 0, st, b);
        VNET_LAUNCH_CHECK();
    }
    return VNET_OK;
}

size_t vnet_wgrad_ws_bytes(int ks, int kx, int stride, int Cin, int Cout, int B, int Do, int Ho, int Wo) {
    if (kx == 0) kx = ks;
    WgradPlan p = plan_wgrad(ks, kx, stride, Cin, Cout, B, Do, Ho, Wo);
    return (size_t)p.nsplit * ks * ks * kx * round_up(Cin, 16) * round_up(Cout, 16) * sizeof(float);
}

int vnet_conv_wgrad(int ks, int kx, int stride, const float* x0, int C0, const float* x1, int C1,
                    const float* dy, int Cout, float* dw,
                    int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                    void* ws, size_t ws_bytes, void* stream) {
    if (!x0 || !dy || !dw || C0 <= 0 || Cout <= 0 || B <= 0 || C1 < 0 || (C1 > 0 && !x1)) return VNET_E_BADARG;
    if (Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return VNET_E_BADARG;
    if (!((ks == 5 && stride == 1) || (ks == 2 && stride == 2))) return VNET_E_UNSUPPORTED;
    if (kx == 0) kx = ks;
    if (kx != ks && !(ks == 5 && kx == 1 && round_up(Cout, 16) == 16)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.dy = dy; a.Cout = Cout;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.CinP = round_up(a.Cin, 16); a.CoutP = round_up(Cout, 16);
    a.pad = ks == 5 ? 2 : 0; a.padx = (kx - 1) / 2;
    a.vec_in = (C0 % 4 == 0) && (C1 % 4 == 0); a.vec_dy = (Cout % 4 == 0);
    WgradPlan p = plan_wgrad(ks, kx, stride, a.Cin, Cout, B, Do, Ho, Wo);
    a.ncob = p.ncob; a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.nbrick = p.nbrick; a.nsplit = p.nsplit;
    const int T3 = ks * ks * kx;
    const size_t need = (size_t)p.nsplit * T3 * a.CinP * a.CoutP * sizeof(float);
    // one slab and no channel padding: the slab IS dw (TF layout [tap][Cin][Cout]) -> no reduce pass
    const bool direct = p.nsplit == 1 && a.CinP == a.Cin && a.CoutP == Cout;
    if (!direct && (!ws || ws_bytes < need)) return VNET_E_WORKSPACE;
    a.part = direct ? dw : reinterpret_cast<float*>(ws);
    int e;
    if (ks == 5 && kx == 1) {
        e = p.small ? launch_wgrad<5, 1, 4, 8, 8, 1, 4, 1>(a, p, st) : launch_wgrad<5, 1, 4, 4, 16, 1, 4, 1>(a, p, st);
    } else if (ks == 5) {
        if (p.small) {
            e = p.ns == 2 ? launch_wgrad<5, 1, 4, 8, 8, 2, 8>(a, p, st) : launch_wgrad<5, 1, 4, 8, 8, 1, 16>(a, p, st);
        } else {
            e = p.ns == 2 ? launch_wgrad<5, 1, 4, 4, 16, 2, 8>(a, p, st) : launch_wgrad<5, 1, 4, 4, 16, 1, 16>(a, p, st);
        }
    } else {
        if (p.small) {
            e = p.ns == 4 ? launch_wgrad<2, 2, 2, 8, 8, 4, 1>(a, p, st) : p.ns == 2 ? launch_wgrad<2, 2, 2, 8, 8, 2, 1>(a, p, st)
                                                                                     : launch_wgrad<2, 2, 2, 8, 8, 1, 1>(a, p, st);
        } else {
            e = p.ns == 4 ? launch_wgrad<2, 2, 2, 4, 16, 4, 1>(a, p, st) : p.ns == 2 ? launch_wgrad<2, 2, 2, 4, 16, 2, 1>(a, p, st)
                                                                                      : launch_wgrad<2, 2, 2, 4, 16, 1, 1>(a, p, st);
        }
    }
    if (e) return e;
    if (direct) return VNET_OK;
    launch_wgrad_reduce(a.part, p.nsplit, T3, a.CinP, a.CoutP, a.Cin, Cout, dw, st);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

size_t vnet_wgrad_bf16_ws_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    WgradPlan p = plan_wgrad(5, 5, 1, Cin, Cout, B, D, H, W, true);
    return (size_t)p.nsplit * 125 * round_up(Cin, 16) * round_up(Cout, 16) * sizeof(float);
}

}  // extern "C"

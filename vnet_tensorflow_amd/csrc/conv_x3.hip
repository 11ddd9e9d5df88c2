// conv_x3.hip -- entry points of the f32x3 convolution family (conv_x3.h): fp32 tensors in and out, fp32 accuracy, bf16 matrix pipe.
#include "conv_x3.h"

namespace {

int x3_grid() { return device_cus() / 8 * 8; }     // one persistent workgroup per CU, a multiple of 8 (items are partitioned by XCD = blockIdx % 8)

}  // namespace

extern "C" {

int vnet_conv_x3_ok(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W) {
    return x3_conv_ok(C0, C1, Cy0, Cy1, B, D, H, W) ? 1 : 0;
}

// rows of the epilogue-statistics buffer [rows][2][Cout]: one per brick (2x8x16; 4x8x8 in volumes 8 wide), or (K-split launches) one per block of the reduce
int vnet_conv_x3_stats_rows(int Cin, int Cout, int B, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const X3Plan p = x3_plan_conv(Cin, 0, Cout, 0, B, D, H, W);
    if (p.nks > 1) {
        if (Cout > 256 || 256 % Cout) return 0;
        const size_t total = (size_t)B * D * H * W * Cout;
        return (int)min((size_t)2048, (total + 255) / 256);
    }
    return x3_conv_bricks(B, D, H, W);
}

size_t vnet_conv_x3_ws_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || (Cin & 15) || (Cout & 15) || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const X3Plan p = x3_plan_conv(Cin, 0, Cout, 0, B, D, H, W);
    return p.nks > 1 ? (size_t)p.nks * B * D * H * W * Cout * sizeof(float) : 0;
}

int vnet_conv_fwd_x3(const float* x0, int C0, const float* x1, int C1, const void* wp, const float* bias,
                     float* y0, int Cy0, float* y1, int Cy1, int B, int D, int H, int W,
                     const float* acc, const float* res, float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (!x0 || !wp || !y0 || C0 <= 0 || Cy0 <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return VNET_E_BADARG;
    if ((C1 > 0 && !x1) || (Cy1 > 0 && !y1) || C1 < 0 || Cy1 < 0) return VNET_E_BADARG;
    if ((C0 & 15) || (C1 & 15) || (Cy0 & 15) || (Cy1 & 15)) return VNET_E_UNSUPPORTED;
    if ((stats || res) && Cy1 > 0) return VNET_E_BADARG;
    if (acc && acc != y0 && Cy1 > 0) return VNET_E_BADARG;
    if ((long long)D * H * W * (C0 > C1 ? C0 : C1) >= (1ll << 31)) return VNET_E_UNSUPPORTED;
    ConvArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1;
    a.wp = reinterpret_cast<const float4*>(wp); a.bias = bias;
    a.y0 = y0; a.y1 = y1; a.Cy0 = Cy0; a.Cy1 = Cy1; a.Cout = Cy0 + Cy1;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W; a.Do = D; a.Ho = H; a.Wo = W;
    a.nchunks = a.Cin / 16; a.CQ = a.nchunks * 4; a.CoutP = a.Cout;
    a.nbz = x3_conv_nbz(D, W); a.nby = ceil_div(H, X3_TY); a.nbx = ceil_div(W, X3_TX);
    a.pad = 2; a.padx = 2; a.vec_in = 1; a.vec_out = 1;
    a.accum = acc ? 1 : 0; a.accsrc = (acc && acc != y0) ? acc : nullptr;
    a.res = res; a.stats = stats;
    // (the plan is that of the whole problem: the same whether the channels come from one source or two, go to one output or two)
    const X3Plan p = x3_plan_conv(a.Cin, 0, a.Cout, 0, B, D, H, W);
    a.nz = p.nks; a.cps = p.cps ? p.cps : a.nchunks;
    const size_t nvox = (size_t)B * D * H * W;
    if (p.nks > 1) {
        const size_t need = (size_t)p.nks * nvox * a.Cout * sizeof(float);
        if (!ws || ws_bytes < need) return VNET_E_WORKSPACE;
        if (stats && (a.Cout > 256 || 256 % a.Cout)) return VNET_E_UNSUPPORTED;
        a.part = reinterpret_cast<float*>(ws); a.part_stride = nvox * a.Cout;
    }
    hipStream_t st = (hipStream_t)stream;
    const int grid = x3_grid();
    // two 16-cout blocks per item where the layer has them and the pairs are still one round of the chip (a pair may straddle
    // y0 / y1 -- each block finds its own destination: the backward-data launch of a 2C -> C layer with C = 16).  Round 6: from 2 x grid single-block items on (was 4 x): 32^3 64->64 0.149 -> 0.140 ms, 128->64 0.275 -> 0.263
    const bool w8 = W == 8;                            // the narrow brick (X3G<true>): one cout block per item
    const bool nb2 = !w8 && (a.Cout % 32 == 0) && p.items * p.nks >= 2 * (long)grid && tuning().x3_nb2 != 0;
#define VNET_X3_GO(STATSV, NBV, W8V)                                                     \
    {                                                                                    \
        auto k = conv5_x3_kernel<STATSV, NBV, W8V>;                                      \
        static unsigned long long attr_done = 0;                                         \
        if (int ae = ensure_lds(k, X3G<W8V>::LDS, attr_done)) return ae;                 \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), X3G<W8V>::LDS, st, a);              \
    }
    if (stats && p.nks == 1) { if (w8) VNET_X3_GO(true, 1, true) else if (nb2) VNET_X3_GO(true, 2, false) else VNET_X3_GO(true, 1, false) }
    else { if (w8) VNET_X3_GO(false, 1, true) else if (nb2) VNET_X3_GO(false, 2, false) else VNET_X3_GO(false, 1, false) }
#undef VNET_X3_GO
    VNET_LAUNCH_CHECK();
    if (p.nks > 1) {
        const size_t total = nvox * a.Cout;
        const int blocks = (int)min((size_t)2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.part, a.part_stride, p.nks, bias,
                           y0, y1, Cy0, Cy1, a.CoutP, nvox, a.accum, a.res, a.stats, a.accsrc);
        VNET_LAUNCH_CHECK();
    }
    return VNET_OK;
}

int vnet_wgrad_x3_ok(int C0, int C1, int Cout, int B, int D, int H, int W) {
    return x3_wgrad_ok(C0, C1, Cout, B, D, H, W) ? 1 : 0;
}

size_t vnet_wgrad_x3_ws_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || (Cin & 15) || (Cout & 15) || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const X3WgradPlan p = x3_plan_wgrad(Cin, Cout, B, D, H, W);
    return (size_t)p.nsplit * 125 * Cin * Cout * sizeof(float);
}

int vnet_conv_wgrad_x3(const float* x0, int C0, const float* x1, int C1, const float* dy, int Cout, float* dw,
                       int B, int D, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!x0 || !dy || !dw || C0 <= 0 || Cout <= 0 || B <= 0 || C1 < 0 || (C1 > 0 && !x1)) return VNET_E_BADARG;
    if (D <= 0 || H <= 0 || W <= 0) return VNET_E_BADARG;
    if ((C0 & 15) || (C1 & 15) || (Cout & 15)) return VNET_E_UNSUPPORTED;
    if ((long long)D * H * W * (C0 > C1 ? C0 : C1) >= (1ll << 31)) return VNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.dy = dy; a.Cout = Cout;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W; a.Do = D; a.Ho = H; a.Wo = W;
    a.CinP = a.Cin; a.CoutP = Cout; a.pad = 2; a.padx = 2; a.vec_in = 1; a.vec_dy = 1;
    const X3WgradPlan p = x3_plan_wgrad(a.Cin, Cout, B, D, H, W);
    a.ncob = Cout / 16; a.nbz = p.nbz; a.nby = p.nby; a.nbx = p.nbx; a.nbrick = p.nbrick; a.nsplit = p.nsplit;
    const size_t need = (size_t)p.nsplit * 125 * a.Cin * Cout * sizeof(float);
    const bool direct = p.nsplit == 1;                 // one slab: the slab IS dw (TF layout [tap][Cin][Cout])
    if (!direct && (!ws || ws_bytes < need)) return VNET_E_WORKSPACE;
    a.part = direct ? dw : reinterpret_cast<float*>(ws);
    if (W == 8) {                                      // the narrow brick (XWG<true>)
        auto k = wgrad5_x3_kernel<true>;
        static unsigned long long attr_done = 0;
        if (int ae = ensure_lds(k, XWG<true>::LDS, attr_done)) return ae;
        hipLaunchKernelGGL(k, dim3(p.nsplit, p.nblk), dim3(512), XWG<true>::LDS, st, a);
    } else {
        auto k = wgrad5_x3_kernel<false>;
        static unsigned long long attr_done = 0;
        if (int ae = ensure_lds(k, XW_LDS, attr_done)) return ae;
        hipLaunchKernelGGL(k, dim3(p.nsplit, p.nblk), dim3(512), XW_LDS, st, a);
    }
    VNET_LAUNCH_CHECK();
    if (direct) return VNET_OK;
    launch_wgrad_reduce(a.part, p.nsplit, 125, a.CinP, a.CoutP, a.Cin, Cout, dw, st);
    VNET_LAUNCH_CHECK();
    return VNET_OK;
}

#ifdef VNET_STAMPS
// experiment build only (profiles/x3_stamps.sh): where conv5_x3_kernel writes its s_memtime stamps (4 steps x 8 waves x 12 int64)
int vnet_debug_set_stamps_x3(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)); }
#endif

}  // extern "C"

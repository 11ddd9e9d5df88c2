/* vnet_hip.h -- C ABI of libvnet_hip.so: the MI355X (gfx950) kernels behind the V-Net hot path.
 * The reference (jackyko1991/vnet-tensorflow) has no native ABI: its hot path is a list of stock TensorFlow-1.15 ops reached from
 * Python.  Each entry point replaces the TF op named in its comment at the cited reference call site; a maintainer binds them with
 * ctypes; vnet_tensorflow_amd/_lib.py is that binding.
 * Conventions
 *   - tensors are float32, channels-last (NDHWC), contiguous; labels int32; `*_b16` entry points take bf16 TENSORS (2-byte elements).
 *   - every pointer is a DEVICE pointer owned by the caller (incl. workspace `ws`); the library allocates nothing and keeps no
 *     per-call state (two opt-in, caller-driven exceptions: the per-stream queue of deferred filter-gradient reduces between
 *     vnet_wgrad_defer(1, stream) and vnet_wgrad_flush(stream), and the tuning options of vnet_set_option); all work is enqueued
 *     on `stream` (hipStream_t), no hidden synchronisation -> safe next to RCCL on another stream and inside a hipGraph capture.
 *   - return value: 0, a negative VNET_E_* code for argument errors, or a positive hipError_t.  Nothing throws, nothing exits.
 *   - round 5 retired round 2's bf16 shadows of fp32 tensors (`*_x16`, `vnet_conv_*_bf16`): the storage mode supersedes them. */
#ifndef VNET_HIP_H
#define VNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNET_OK            0
#define VNET_E_BADARG     -1   /* null pointer / non-positive dimension                  */
#define VNET_E_UNSUPPORTED -2  /* kernel size / stride / channel multiple not instantiated */
#define VNET_E_WORKSPACE  -3   /* ws too small: query vnet_conv_ws_bytes / vnet_wgrad_ws_bytes */

/* activation kinds for vnet_bn_act_* (networks.py:239-244) */
#define VNET_ACT_NONE  0
#define VNET_ACT_RELU  1       /* tf.nn.relu                                              */
#define VNET_ACT_PRELU 2       /* layers2.py:97-99                                        */
#define VNET_ACT_LRELU 3       /* tf.nn.leaky_relu (alpha 0.2)                            */

/* weight packing modes for vnet_pack_weights */
#define VNET_PACK_FWD  0       /* conv forward:        wp[t][ci/4][co][ci%4] = w[t][ci][co]                 */
#define VNET_PACK_BWD  1       /* conv backward-data:  wp[t][co/4][ci][co%4] = w[T-1-t][ci][co]             */
#define VNET_PACK_UP   2       /* 2x2x2 transposed:    wp[0][ci/4][a*O+o][ci%4] = w[a][o][ci], w=[8][O][I]  */

/* bf16 filter images for vnet_conv_fwd_b16: [cin chunk 16][tap][cout block 32][cin half][32 cout][8 cin] */
#define VNET_PACK_FWD_BF16 3   /* conv forward, operands rounded to bf16 (RNE)                              */
#define VNET_PACK_BWD_BF16 4   /* conv backward-data (flipped taps, cin<->cout), bf16                       */
/* vnet_pack_weights_batched only: BOTH images (forward, backward-data) of a filter from one read of w; descriptor
 * {w, wp_fwd, mode, T, I, O, wp_bwd, 0}, I and O multiples of 32 */
#define VNET_PACK_BOTH 6       /* the fp32 pair */
#define VNET_PACK_BOTH_BF16 5  /* the bf16 pair of a 5^3 filter */

/* f32x3 filter images for vnet_conv_fwd_x3: every weight split exactly into three bf16 pieces (h, m, l),
 * [k chunk 16][tap pair 63][n block 16][piece 3][64 lanes][8 k] (csrc/conv_x3.h, x3_pack.h: x3_pair_tap) */
#define VNET_PACK_FWD_X3 7     /* conv forward (k = ci, n = co)                                              */
#define VNET_PACK_BWD_X3 8     /* conv backward-data (flipped taps, k = co, n = ci)                          */
#define VNET_PACK_BOTH_X3 9    /* vnet_pack_weights_batched only, like VNET_PACK_BOTH: both f32x3 images of a 5^3 filter  */

/* or-ed into VNET_PACK_FWD / _BWD / _UP: the fp32 image holds the filter ROUNDED to bf16 (bf16-storage mode of the 2^3 convs) */
#define VNET_PACK_ROUND_BF16 16

/* loss kinds for vnet_softmax_dice_* (model.py:495-558), and the flags or-ed into them */
#define VNET_LOSS_SORENSEN 0
#define VNET_LOSS_JACCARD  1
#define VNET_LOSS_XENT     2
#define VNET_LOSS_WEIGHTED 16  /* weighted_* variants (model.py:70-75 / 87-92)            */
#define VNET_LOSS_MIXED    32  /* mixed_* = dice + Alpha * xent (model.py:524-556)        */

const char* vnet_version(void);

/* Tuning options: read ONCE from the environment (VNET_<NAME>, first use), never on a launch path; afterwards only through
 * vnet_set_option(name, value) -> previous value (NaN: unknown name).  Names: WGRAD_ZS, WGRAD_RR, CONV_IN4, WGRAD_GROUP_ROUNDS,
 * WGRAD_GROUP_DEBUG, BF16_DEEP, BF16_DEEP_TARGET, BF16_C16PP, F32_SMALL, X3_NB2 ("Switches"). */
double vnet_set_option(const char* name, double value);
double vnet_get_option(const char* name);

/* ---- weight repacking: a TF filter (layers2.py:60 `weights`, DHWIO [taps][I][O]) re-laid into the MFMA-fragment order the conv
 * kernels stream; output size in floats = vnet_packed_weight_floats(mode, taps, I, O). */
size_t vnet_packed_weight_floats(int mode, int taps, int I, int O);
int vnet_pack_weights(int mode, const float* w, float* wp, int taps, int I, int O, void* stream);
/* Same for every filter of a network in ONE launch (called after each optimiser step): `descs_dev` is a DEVICE
 * array of n records of 8 int64 {w ptr, wp ptr, mode, taps, I, O, CQ, NP} (CQ, NP from vnet_packed_dims). */
int vnet_packed_dims(int mode, int taps, int I, int O, int* CQ, int* NP);
int vnet_pack_weights_batched(const void* descs_dev, int n, void* stream);

/* ---- N-D convolution: tf.nn.convolution(x, w, 'SAME', strides) + b (layers2.py:63) and, with up=1,
 * tf.nn.conv3d_transpose(x, w, output_shape, [1,2,2,2,1], 'SAME') + b (layers2.py:73).  Also backward-data (VNET_PACK_BWD weights;
 * the 2^3 down/up pair are each other's backward-data).
 *   ks/stride : (5,1) or (2,2); up=1 with (2,2) = transposed conv.  kx = kernel extent along x (0 = ks); (ks=5,kx=1) is the 5x5x1
 *               conv on the x-im2col'ed 1-channel input (vnet_tile_im2col_x): 25 taps x 16 virtual channels.
 *   x0,C0,x1,C1 : input = channel-concat of two NDHWC tensors (tf.concat, networks.py:325, never materialised); x1 may be NULL.
 *   wp, bias  : packed weights (vnet_pack_weights) for Cin=C0+C1 -> Cout=Cy0+Cy1; bias [Cout] or NULL.
 *   y0,Cy0,y1,Cy1 : output split over two NDHWC tensors along channels (backward-data of a concat); y1 may be NULL.
 *   B, Di,Hi,Wi / Do,Ho,Wo : input / output spatial dims (SAME: ceil(in/stride); up: skip shape).
 *   ws        : split-K partials, >= vnet_conv_ws_bytes(...) bytes (may be NULL if 0). */
size_t vnet_conv_ws_bytes(int ks, int kx, int stride, int up, int Cin, int Cout, int B, int Do, int Ho, int Wo);
int vnet_conv_fwd(int ks, int kx, int stride, int up, const float* x0, int C0, const float* x1, int C1, const float* wp,
    const float* bias, float* y0, int Cy0, float* y1, int Cy1, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, void* ws,
    size_t ws_bytes, void* stream);

/* y += conv(x): backward-data of a tensor with TWO consumers (skip connection networks.py:276,325; block input networks.py:314-318)
 * adds the second gradient in the epilogue instead of an autodiff add kernel (tf.add_n, model.py:660).  Arguments as vnet_conv_fwd. */
int vnet_conv_fwd_acc(int ks, int kx, int stride, int up, const float* x0, int C0, const float* x1, int C1, const float* wp,
    const float* bias, float* y0, int Cy0, float* y1, int Cy1, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, void* ws,
    size_t ws_bytes, void* stream);

/* Batch-norm statistics in the convolution epilogue (networks.py:316-319: conv, then tf.layers.batch_normalization of conv [+ res]):
 * the launch also writes per-workgroup partial sums of v = y (+ res) and v^2 into stats[rows][2][Cout]; vnet_bn_finalize_partial
 * turns them into mean / invstd (+ moving averages).  rows = vnet_conv_stats_rows(...) (0: this launch cannot -- transposed conv,
 * split output, Cout % 4 != 0, split-K with 256 % Cout != 0 -- use vnet_bn_stats).  One output tensor y; res NULL or like y. */
int vnet_conv_stats_rows(int ks, int kx, int stride, int up, int Cin, int Cy0, int Cy1, int B, int Do, int Ho, int Wo);
int vnet_conv_stats_from_reduce(int ks, int kx, int stride, int Cin, int Cout, int B, int Do, int Ho, int Wo);
    /* 1: split-K launch, the reduce kernel produces them */
int vnet_conv_fwd_stats(int ks, int kx, int stride, const float* x0, int C0, const float* x1, int C1, const float* wp,
    const float* bias, float* y, int Cout, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, const float* res, float* stats,
    void* ws, size_t ws_bytes, void* stream);

/* ---- f32x3 (round 5): the 5^3 stride-1 convolution in fp32 accuracy on the bf16 matrix pipe (layers2.py:59-63 from
 * networks.py:316,333,346; Conv3DBackpropInput, model.py:660).  fp32 tensors; every operand is split EXACTLY into three bf16 pieces
 * and a product is the fp32 sum of six bf16 products (hh, hm, mh, mm, hl, lh; dropped terms < 2^-24 of the product).  Not
 * bit-identical to the fp32 MFMA kernels; held to the same 2e-6 against the fp64 oracle.  wp: VNET_PACK_FWD_X3 (forward) /
 * VNET_PACK_BWD_X3 (backward-data: C0 / Cy0, Cy1 are the backward problem's).  Channel counts % 16 (else VNET_E_UNSUPPORTED);
 * vnet_conv_x3_ok: the kernel is the better choice for the shape (rows of >= 16 voxels -- 2 x 8 x 16 bricks -- or, round 6, volumes
 * exactly 8 wide -- 4 x 8 x 8 bricks; any other width runs on the wide brick with idle columns).  acc: NULL, y0 (y0 += conv) or another tensor of y0's shape
 * (Cy1 == 0) added out of place.  res / stats as vnet_conv_fwd_stats, rows = vnet_conv_x3_stats_rows.  ws >= vnet_conv_x3_ws_bytes. */
int vnet_conv_x3_ok(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W);
int vnet_conv_x3_stats_rows(int Cin, int Cout, int B, int D, int H, int W);
size_t vnet_conv_x3_ws_bytes(int Cin, int Cout, int B, int D, int H, int W);
    /* > 0: the deep levels split their channel chunks over workgroups (partial slabs + reduce) */
int vnet_conv_fwd_x3(const float* x0, int C0, const float* x1, int C1, const void* wp, const float* bias, float* y0, int Cy0,
    float* y1, int Cy1, int B, int D, int H, int W, const float* acc, const float* res, float* stats, void* ws, size_t ws_bytes,
    void* stream);

/* Filter gradient of the same convolution (Conv3DBackpropFilterV2 behind model.py:660) with the six-product arithmetic:
 * dw [125][C0 + C1][Cout] (TF layout).  Channel counts multiples of 16; ws >= vnet_wgrad_x3_ws_bytes (partial slabs, reduced by
 * this call or -- between vnet_wgrad_defer(1) and vnet_wgrad_flush -- by the flush). */
int vnet_wgrad_x3_ok(int C0, int C1, int Cout, int B, int D, int H, int W);
size_t vnet_wgrad_x3_ws_bytes(int Cin, int Cout, int B, int D, int H, int W);
int vnet_conv_wgrad_x3(const float* x0, int C0, const float* x1, int C1, const float* dy, int Cout, float* dw, int B, int D, int H,
    int W, void* ws, size_t ws_bytes, void* stream);

/* workspace of vnet_conv_wgrad_b16 (bf16 storage, below): partial slabs of the split over bricks */
size_t vnet_wgrad_bf16_ws_bytes(int Cin, int Cout, int B, int D, int H, int W);

/* Deferred reduces of the filter-gradient slabs, PER STREAM.  After vnet_wgrad_defer(1, stream) the filter-gradient entry points
 * launching on `stream` leave their partial slabs in the caller's workspace (one per layer, untouched until the flush) and queue
 * the reduce; vnet_wgrad_flush(stream) runs that stream's queue in one launch (model.py:660-666: gradients are complete when
 * compute_gradients returns).  Returns previous setting / queue length / status.  Bit-identical to the per-layer reduce. */
int vnet_wgrad_defer(int on, void* stream);
int vnet_wgrad_pending(void* stream);
int vnet_wgrad_flush(void* stream);

/* ---- convolution filter gradient (the Conv3DBackpropFilterV2 autodiff builds at model.py:660)
 *   dw[t][ci][co] = sum_v x[v*stride + t - pad][ci] * dy[v][co]     (TF layout, unpadded)
 * x is the (possibly two-source) forward input, dy the gradient at the conv output [B,Do,Ho,Wo,Cout].
 * For the transposed 2^3 conv call it with x := dy_fine, dy := x_coarse (ks=2,stride=2). */
size_t vnet_wgrad_ws_bytes(int ks, int kx, int stride, int Cin, int Cout, int B, int Do, int Ho, int Wo);
int vnet_conv_wgrad(int ks, int kx, int stride, const float* x0, int C0, const float* x1, int C1, const float* dy, int Cout,
    float* dw, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, void* ws, size_t ws_bytes, void* stream);

/* ---- single-modality input block (networks.py:254-259 tile + BN feeding level-1 conv_1, networks.py:316): every channel of that
 * conv's input is an affine function of one image, so it collapses to a 5x5x1 conv over 16 virtual channels (x-im2col of image and
 * inside-indicator): vnet_tile_im2col_x img [B,D,H,W,1] -> xv [..,16]; vnet_input_conv_fold w [125][C][O] + BN coefficients ->
 * wv [25][16][O]; forward = vnet_conv_fwd(5, kx=1, xv); G = vnet_conv_wgrad(5, kx=1, xv, dy); vnet_input_conv_grads G -> dw
 * [125][C][O] + the conv-path parts of dgamma / dbeta (accumulate=1 adds). */
int vnet_tile_im2col_x(const float* img, float* xv, int B, int D, int H, int W, void* stream);
int vnet_input_conv_fold(const float* w, const float* gamma, const float* beta, const float* mean, const float* invstd, float* wv,
    int C, int O, void* stream);
int vnet_input_conv_grads(const float* G, const float* w, const float* gamma, const float* beta, const float* mean,
    const float* invstd, float* dw, float* dgamma, float* dbeta, int C, int O, int accumulate, void* stream);

/* Round 6: the same folded block WITHOUT the x-im2col tensor and without the matrix cores -- packed fp32 FMAs on the vector pipe straight
 * from the 1-channel image (csrc/input_block.hip: 8.4 GF of real work instead of 26.8 GF of 5x5x1 MFMA work; every product an exact
 * fp32 FMA; the indicator channel is separable and costs no multiplications: vnet_input_conv_fold_border pre-sums its x taps per
 * x class of a voxel).  O = 8 or 16 (vnet_input_conv_direct_ok), wv from vnet_input_conv_fold.
 *   forward : y = conv(fold) + bias; stats (optional) [rows = vnet_input_conv_direct_stats_rows][2][O] partial batch-norm sums of y (+ res)
 *   gradient: G [25][16][O] in the layout vnet_input_conv_grads reads; ws >= vnet_input_wgrad_direct_slabs(..) * 25 * 16 * O floats */
int vnet_input_conv_direct_ok(int O, int B, int D, int H, int W);
int vnet_input_conv_direct_stats_rows(int B, int D, int H, int W);
int vnet_input_conv_fold_border(const float* wv, int O, float* wbc /* [9][25][O] */, float* cbc /* [9][O] */, void* stream);
int vnet_input_conv_direct_fwd(const float* img, const float* wv, const float* wbc, const float* cbc, const float* bias, const float* res,
    float* y, float* stats, int O, int B, int D, int H, int W, void* stream);
int vnet_input_wgrad_direct_slabs(int B, int D, int H, int W);
int vnet_input_wgrad_direct(const float* img, const float* dy, float* G, int O, int B, int D, int H, int W, void* ws, size_t ws_bytes,
    void* stream);

/* ---- 1x1x1 output head, networks.py:298-303 `convolution(x,[1,1,1,C,K])` (K <= 8) ---------- */
int vnet_head_fwd(const float* x, const float* w, const float* bias, float* y, int64_t M, int C, int K, void* stream);
int vnet_head_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int64_t M, int C, int K,
    void* ws, size_t ws_bytes, void* stream);
size_t vnet_head_ws_bytes(int C, int K);

/* ---- column sums: db = sum over voxels of dy (bias gradient of layers2.py:61) --------------- */
size_t vnet_colsum_ws_bytes(int C);
int vnet_colsum(const float* x, float* out, int64_t M, int C, void* ws, size_t ws_bytes, void* stream);

/* ---- train-mode batch-norm + residual + activation: tf.layers.batch_normalization(momentum=.99, epsilon=.001, training=True)
 * (networks.py:259,265,279,293,303,319,334-337,347,358,361) fused with the residual add in front (networks.py:318,336,360), the
 * tf.tile of the 1-channel input (bcast=1: x has 1 channel, broadcast to C) and the activation behind it (layers2.py:97-99).
 *   s = x (+ r);  mean/var over M = B*D*H*W rows (biased);  y = act(gamma*(s-mean)*invstd+beta)
 * stats: writes mean[C], invstd[C]; moving -= (moving - batch) * (1 - momentum) (moving_* may be NULL). */
size_t vnet_bn_ws_bytes(int C);
int vnet_bn_stats(const float* x, const float* r, int bcast, int64_t M, int C, float eps, float momentum, float* mean,
    float* invstd, float* moving_mean, float* moving_var, void* ws, size_t ws_bytes, void* stream);
int vnet_bn_act_fwd(const float* x, const float* r, int bcast, int64_t M, int C, const float* mean, const float* invstd,
    const float* gamma, const float* beta, int act, const float* alpha, float* y, void* stream);
/* backward: pass 1 reduces dgamma,dbeta,dalpha; pass 2 writes ds (gradient w.r.t. s = x + r;
 * the same tensor is the gradient of both x and r).  bcast=1: ds has C channels (caller sums). */
int vnet_bn_act_bwd(const float* dy, const float* x, const float* r, int bcast, int64_t M, int C, const float* mean,
    const float* invstd, const float* gamma, const float* beta, int act, const float* alpha, float* dgamma, float* dbeta,
    float* dalpha, float* ds, void* ws, size_t ws_bytes, void* stream);

/* Cross-replica ("sync") batch-norm, SURVEY 8(e)(ii): the host all-reduces small vectors between the pieces --
 * vnet_bn_moments -> sums[2C] doubles [all-reduce]; vnet_bn_finalize -> mean / invstd from global sums and M_total rows;
 * vnet_bn_act_bwd_reduce -> local dgamma/dbeta/dalpha [all-reduce copies]; vnet_bn_act_bwd_apply -> ds from the GLOBAL sums.
 * vnet_bn_stats == moments + finalize (M_total = M); vnet_bn_act_bwd == reduce + apply.  vnet_bn_finalize_partial: the same from
 * `rows` partial rows [2][C] of (sum, sum of squares), e.g. a convolution's epilogue (vnet_conv_fwd_stats). */
int vnet_bn_finalize_partial(const float* partial, int rows, int C, double M_total, float eps, float momentum, float* mean,
    float* invstd, float* moving_mean, float* moving_var, void* stream);
int vnet_bn_moments(const float* x, const float* r, int bcast, int64_t M, int C, double* sums, void* ws, size_t ws_bytes,
    void* stream);
int vnet_bn_finalize(const double* sums, double M_total, int C, float eps, float momentum, float* mean, float* invstd,
    float* moving_mean, float* moving_var, void* stream);
int vnet_bn_act_bwd_reduce(const float* dy, const float* x, const float* r, int bcast, int64_t M, int C, const float* mean,
    const float* invstd, const float* gamma, const float* beta, int act, const float* alpha, float* dgamma, float* dbeta,
    float* dalpha, void* ws, size_t ws_bytes, void* stream);
int vnet_bn_act_bwd_apply(const float* dy, const float* x, const float* r, int bcast, int64_t M, int C,
                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                          int act, const float* alpha, const float* sum_dz, const float* sum_dz_xhat, double M_total,
                          const float* xhat_coef /* NULL, or [C]: ds += xhat * xhat_coef (batch-norm chains) */,
                          float* ds, void* stream);

/* Batch-norm CHAINS of the decoder in closed form on ONE tensor x (the convolution output):
 *   kind 0 (networks.py:333-337): y1 = BN1(x); y2 = BN2(y1); out = act(BN3(y1 + y2));  kind 1 (networks.py:358-361): r = BNa(x);
 *   out = act(BNb(x + r)) [(g1,b1) := BNa, (g2,b2) := BNb, the *3 arguments ignored].
 * Every tensor of a chain is per-channel affine in xhat = (x - mean) * invstd with exactly known batch moments, so
 * out = act(ceff * xhat + deff): vnet_bn_stats(x) + vnet_bn_chain_coef_fwd (also the moving-average updates of the derived layers,
 * mm2/mv2, mm3/mv3, NULL to skip) + vnet_bn_act_fwd(gamma := ceff, beta := deff).  Backward: vnet_bn_act_bwd_reduce(ceff, deff) ->
 * dC, dD; vnet_bn_chain_coef_bwd -> every gamma / beta gradient of the chain and xhat_coef (ceff depends on the batch variance);
 * vnet_bn_act_bwd_apply(..., xhat_coef) -> ds.  With cross-replica statistics dC_global is the all-reduced dC. */
int vnet_bn_chain_coef_fwd(int kind, int C, float eps, float momentum, const float* mean, const float* invstd, const float* g1,
    const float* b1, const float* g2, const float* b2, const float* g3, const float* b3, float* ceff, float* deff, float* mm2,
    float* mv2, float* mm3, float* mv3, void* stream);
int vnet_bn_chain_coef_bwd(int kind, int C, float eps, double M_total, const float* mean, const float* invstd, const float* g1,
    const float* g2, const float* g3, const float* dC_local, const float* dD_local, const float* dC_global, float* dg1, float* db1,
    float* dg2, float* db2, float* dg3, float* db3, float* xhat_coef, void* stream);

/* ---- stand-alone activation, layers2.py:97-99 prelu / tf.nn.relu / tf.nn.leaky_relu (in the networks it is fused into vnet_bn_act_*) */
int vnet_act_fwd(const float* x, int64_t M, int C, int act, const float* alpha, float* y, void* stream);
int vnet_act_bwd(const float* dy, const float* x, int64_t M, int C, int act, const float* alpha, float* dalpha, float* dx, void* ws,
    size_t ws_bytes, void* stream);

/* ---- fused softmax + soft-Dice / cross-entropy loss head: tf.nn.softmax (model.py:447) + tf.one_hot (model.py:474-477) + dice_coe
 * (model.py:26-85) + the loss switch (model.py:495-558).  logits [B,V,K], labels int32 [B,V].
 *   loss_kind VNET_LOSS_* | flags; weights [K] or NULL; alpha = Loss.Alpha; softmax_out optional [B,V,K]; pred_out optional int64
 *   argmax (model.py:567-568); loss_out / dice_out (optional) device scalars; coef [B][K][2] + 1 floats kept for the backward pass.
 * backward: dlogits = d loss / d logits * (*gscale)  (gscale device scalar, may be NULL = 1). */
size_t vnet_loss_ws_bytes(int B, int K);
int vnet_softmax_dice_fwd(const float* logits, const int32_t* labels, int B, int64_t V, int K, int loss_kind, const float* weights,
    float alpha, float smooth, float* softmax_out, int64_t* pred_out, float* loss_out, float* dice_out, float* coef, void* ws,
    size_t ws_bytes, void* stream);
int vnet_softmax_dice_bwd(const float* logits, const int32_t* labels, int B, int64_t V, int K, int loss_kind, const float* weights,
    float alpha, const float* coef, const float* gscale, float* dlogits, void* stream);

/* ---- stand-alone dice_coe(output, target, loss_type, axis=(1,2,3), weights, smooth) (model.py:26-85)
 * on probability / one-hot tensors [B,V,K]; ws >= vnet_loss_ws_bytes(B,K) + 4.  coef as above. */
int vnet_dice_coe_fwd(const float* output, const float* target, int B, int64_t V, int K, int jaccard, const float* weights,
    float smooth, float* dice_out, float* coef, void* ws, size_t ws_bytes, void* stream);
int vnet_dice_coe_bwd(const float* output, const float* target, int B, int64_t V, int K, int jaccard, const float* coef,
    const float* gscale, float* doutput, void* stream);

/* ---- dropout, tf.nn.dropout(x, rate) (networks.py:321,339,349,363): counter-based RNG ------- */
int vnet_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float rate, uint64_t seed, void* stream);
int vnet_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float rate, void* stream);

/* ---- optimiser apply ops (model.py:649-656), fused over one flat fp32 parameter buffer.  TF1 AdamOptimizer: lr_t =
 * lr*sqrt(1-b2^t)/(1-b1^t) (passed in: the host computes the scalar schedule, model.py:641-644); m,v update; p -= lr_t*m/(sqrt(v)+eps).
 * gscale multiplies the gradient first (1/world for the data-parallel mean). */
int vnet_adam_apply(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2, float eps,
    float gscale, void* stream);
int vnet_sgd_apply(float* p, const float* g, int64_t n, float lr, float gscale, void* stream);
int vnet_momentum_apply(float* p, const float* g, float* acc, int64_t n, float lr, float momentum, int nesterov, float gscale,
    void* stream);

/* ---- device-resident step state (model.py:743-748: one sess.run per step).  A hipGraph freezes kernel ARGUMENTS, not memory:
 * `state` = caller-owned 32 device bytes {float lr; float lr_t; u32 pad[2]; u64 step; u64 pad}, written by vnet_step_state_set
 * (one-thread kernel, launched eagerly before each replay; model.py:641-644).  The *_dev optimisers read lr / lr_t from it and
 * the dropout adds step * odd constant to its seed, so a captured graph of the WHOLE step replays unchanged. */
int vnet_step_state_set(void* state, float lr, float lr_t, uint64_t step, void* stream);
int vnet_adam_apply_dev(float* p, const float* g, float* m, float* v, int64_t n, const void* state, float beta1, float beta2,
    float eps, float gscale, void* stream);
int vnet_sgd_apply_dev(float* p, const float* g, int64_t n, const void* state, float gscale, void* stream);
int vnet_momentum_apply_dev(float* p, const float* g, float* acc, int64_t n, const void* state, float momentum, int nesterov,
    float gscale, void* stream);
int vnet_dropout_fwd_dev(const float* x, float* y, uint8_t* mask, int64_t n, float rate, uint64_t seed, const void* state,
    void* stream);

/* ---- hard metrics (model.py:588-626): K x K confusion matrix cm[label][prediction] as float64 counts;
 * accuracy, per-class tp/tn/fp/fn, sensitivity, specificity and hard Dice 2tp/(2tp+fp+fn) follow on the host. */
size_t vnet_confusion_ws_bytes(int K);
int vnet_confusion_matrix(const int64_t* pred, const int32_t* labels, int64_t n, int K, double* cm_out, void* ws, size_t ws_bytes,
    void* stream);

/* ---- tf.metrics.auc (model.py:607,613,624; TF default: 200 thresholds, ROC, trapezoid, `prediction > threshold` in float32).
 * One pass: hist_out[0][b] (voxels of class cls) / hist_out[1][b] (others), b = number of thresholds strictly below the prediction
 * (0..T), float64 counts; tp / fp by suffix sums and the AUC on the host (ops.auc_from_hist).  thresholds: DEVICE float32[T]. */
size_t vnet_auc_ws_bytes(int T);
int vnet_auc_histogram(const float* softmax, const int32_t* labels, int64_t n, int K, int cls, const float* thresholds, int T,
    double* hist_out, void* ws, size_t ws_bytes, void* stream);

/* ---- sliding-window accumulation for evaluate (model.py:919-929) ----------------------------- */
int vnet_accumulate_patch(const float* patch, float* vol, float* count, int K, int pz, int py, int px, int z0, int y0, int x0,
    int D, int H, int W, void* stream);

/* ==== bf16-STORAGE mode (`*_b16`): BASELINE config C5 as SURVEY 8(d) states it =============================================
 * "bf16 activations/weights into MFMA, fp32 accumulate, fp32 BN stats and Dice sums".  Activations, skip tensors and their
 * gradients are bf16 NDHWC tensors (`void*`, 2 bytes per element, 16-byte aligned, channel counts multiples of 8 -- and of the
 * form 8 * 2^k for the batch-norm kernels); every kernel computes in fp32 and rounds its output ONCE (round-to-nearest-even,
 * v_cvt_pk_bf16_f32).  Filters, biases, batch-norm parameters / statistics, logits, loss and every parameter gradient stay
 * fp32.  Same reference call sites as the fp32 entry points they mirror (layers2.py:59-99, networks.py:259...361,
 * model.py:660); argument meaning as there unless noted. */

/* network input: fp32 [M][C] -> bf16 [M][Cpad], channels C..Cpad-1 zero (Cpad % 8 == 0): the multi-modality image padded to
 * the 16-byte unit the convolution kernels stage (the filter's packed image is zero-padded to 16 input channels anyway) */
int vnet_cast_bf16(const float* x, void* y16, int64_t M, int C, int Cpad, void* stream);

/* per-channel sum of a bf16 [M][C] tensor in fp32 (the bias gradient sum(dy) of layers2.py:63 / :73 in bf16 storage, outside the
 * networks' closed form); C % 8 == 0, deterministic; ws >= vnet_colsum_b16_ws_bytes(C) */
size_t vnet_colsum_b16_ws_bytes(int C);
int vnet_colsum_b16(const void* x16, float* out, int64_t M, int C, void* ws, size_t ws_bytes, void* stream);

/* 5^3 stride-1 convolution, bf16 in / bf16 out; forward (VNET_PACK_FWD_BF16) and backward-data (VNET_PACK_BWD_BF16); operands bf16,
 * fp32 accumulate (v_mfma_f32_32x32x16_bf16 / 16x16x32), ONE rounding.  acc16: NULL or a bf16 tensor of y0's shape added before the
 * rounding (== y0: in place; else out of place, Cy1 = 0); res16 / stats: statistics of the ROUNDED output (+ res16), rows =
 * vnet_conv_b16_stats_rows; ws >= vnet_conv_b16_ws_bytes.  Few bricks with whole 32-cout blocks / 16-cin chunks (the deep levels)
 * take csrc/conv_deep.h, which sums in another fixed order (option BF16_DEEP = 0: the generic kernels). */
size_t vnet_conv_b16_ws_bytes(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W);
int vnet_conv_b16_stats_rows(int C0, int C1, int Cy0, int Cy1, int B, int D, int H, int W);
int vnet_conv_fwd_b16(const void* x0, int C0, const void* x1, int C1, const void* wp, const float* bias, void* y0, int Cy0,
    void* y1, int Cy1, int B, int D, int H, int W, const void* acc16, const void* res16, float* stats, void* ws, size_t ws_bytes,
    void* stream);
/* The same for a zero-padded source: x16 has Cpad channels of which only the first Cin are non-zero (the cast network input of
 * a multi-modality net, vnet_cast_bf16; the packed filter has Cin input channels).  Cin <= 4, Cpad == 8 and a shape the 16-cout
 * kernel takes: x-im2col while staging (K-channels = 4 x shifts x 4 modalities), 2.5x fewer MFMAs; else == vnet_conv_fwd_b16. */
int vnet_conv_fwd_b16_padded(const void* x16, int Cpad, int Cin, const void* wp, const float* bias, void* y16, int Cout, int B,
    int D, int H, int W, const void* res16, float* stats, void* ws, size_t ws_bytes, void* stream);
/* its filter gradient: x, dy bf16, dw fp32 [125][Cin_dw][Cout] with Cin_dw <= C0 + C1 (the leading input channels: a
 * zero-padded network input); ws >= vnet_wgrad_bf16_ws_bytes(C0 + C1, ...) */
int vnet_conv_wgrad_b16(const void* x0, int C0, const void* x1, int C1, const void* dy, int Cout, float* dw, int Cin_dw, int B,
    int D, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* Round 4: the filter gradients of SEVERAL layers in one launch (the independent gradient nodes tf.gradients creates for every
 * layers2.py:59-63 convolution, model.py:660): the workgroups share the CUs according to each layer's work.  jobs[i] = arguments
 * of vnet_conv_wgrad_b16 for layer i (tensors valid until the launch has run); a shape the grouped kernels do not take runs as its
 * own launch.  Honours vnet_wgrad_defer.  A layer's result depends on how the group splits it, never on other layers' data.
 * Option WGRAD_GROUP_ROUNDS (default 2): workgroups per CU the plan aims at; 0 = every job on its own. */
typedef struct vnet_wgrad_job {
    const void* x0; const void* x1; const void* dy; float* dw; void* ws; size_t ws_bytes;
    int C0, C1, Cout, Cin_dw, B, D, H, W;
    int ks;     /* 0 or 5: the 5^3 stride-1 convolution (arguments of vnet_conv_wgrad_b16); 2: the 2^3 stride-2 convolution -- x0 = the
                 * FINE tensor [B,D,H,W,C0], dy = the COARSE tensor [B,ceil(D/2),ceil(H/2),ceil(W/2),Cout], dw [8][C0][Cout]
                 * (arguments of vnet_conv2_wgrad_b16; C1 = 0, Cin_dw = C0) */
} vnet_wgrad_job;
size_t vnet_wgrad_job_bytes(void);   /* sizeof(vnet_wgrad_job): a binding checks its own layout against it */
int vnet_conv_wgrad_b16_group(const vnet_wgrad_job* jobs, int n, void* stream);
/* 2^3 stride-2 convolution (up = 0) / 2^3 transposed convolution (up = 1), bf16 in / bf16 out.  wp: the fp32 packed image of
 * the bf16-ROUNDED filter, vnet_pack_weights(VNET_PACK_FWD | VNET_PACK_ROUND_BF16, 8, Cin, Cout) resp. VNET_PACK_UP | ...;
 * accum: y += result (one rounding of the sum); stats (up = 0 only): rows = vnet_conv_stats_rows(2, 0, 2, 0, ...);
 * ws >= vnet_conv_ws_bytes(2, 0, 2, up, ...).  Channel counts multiples of 4. */
int vnet_conv2_fwd_b16(int up, const void* x, int Cin, const float* wp, const float* bias, void* y, int Cout, int B, int Di, int Hi,
    int Wi, int Do, int Ho, int Wo, int accum, float* stats, void* ws, size_t ws_bytes, void* stream);
/* filter gradient of the 2^3 stride-2 convolution: x = fine tensor [B,Di,Hi,Wi,Cin], dy = coarse tensor [B,Do,Ho,Wo,Cout],
 * dw fp32 [8][Cin][Cout]; ws >= vnet_wgrad_ws_bytes(2, 0, 2, ...) */
int vnet_conv2_wgrad_b16(const void* x, int Cin, const void* dy, int Cout, float* dw, int B, int Di, int Hi, int Wi, int Do, int Ho,
    int Wo, void* ws, size_t ws_bytes, void* stream);

/* The same pair WITHOUT an LDS tile (csrc/conv2_b16.hip; V-Net levels 1-2: vnet_conv2_direct_ok(Cf in {16,32}, Cc in {32,64})).
 * w: fp32 filter in TF layout [8][Cf][Cc] for BOTH layers, rounded to bf16 in the kernel.  down = 1: coarse out = conv(fine in) +
 * bias[Cc], optional statistics rows (vnet_conv2_direct_stats_rows); down = 0: fine out (+)= transposed conv(coarse in) + bias[Cf]. */
int vnet_conv2_direct_ok(int Cf, int Cc);
int vnet_conv2_direct_stats_rows(int Cf, int Cc, int B, int Dc, int Hc, int Wc);
int vnet_conv2_direct_b16(int down, const void* in, void* out, const float* w, const float* bias, int Cf, int Cc, int B, int Df,
    int Hf, int Wf, int Dc, int Hc, int Wc, int accum, float* stats, void* stream);
/* fp32 twin (the reference's arithmetic, v_mfma_f32_16x16x4_f32): same arguments on float tensors, same widths; replaces
 * vnet_conv_fwd(ks = 2, stride = 2, up = 0 | 1) / vnet_conv_fwd_acc / vnet_conv_fwd_stats there (layers2.py:78-94). */
int vnet_conv2_direct_f32(int down, const float* in, float* out, const float* w, const float* bias, int Cf, int Cc, int B, int Df,
    int Hf, int Wf, int Dc, int Hc, int Wc, int accum, float* stats, void* stream);

/* batch-norm (+ residual, + activation) on bf16 tensors; statistics and parameter gradients fp32 (partial rows summed in float64).
 * bcast = 1: x is the fp32 1-channel image [M] broadcast to C channels (tf.tile, networks.py:258), y still bf16 [M][C]. */
int vnet_bn_stats_b16(const void* x16, const void* r16, int64_t M, int C, float eps, float momentum, float* mean, float* invstd,
    float* moving_mean, float* moving_var, void* ws, size_t ws_bytes, void* stream);
int vnet_bn_moments_b16(const void* x16, const void* r16, int64_t M, int C, double* sums, void* ws, size_t ws_bytes, void* stream);
int vnet_bn_act_fwd_b16(const void* x, const void* r16, int bcast, int64_t M, int C, const float* mean, const float* invstd,
    const float* gamma, const float* beta, int act, const float* alpha, void* y16, void* stream);
int vnet_bn_act_bwd_reduce_b16(const void* dy16, const void* x, const void* r16, int bcast, int64_t M, int C, const float* mean,
    const float* invstd, const float* gamma, const float* beta, int act, const float* alpha, float* dgamma, float* dbeta,
    float* dalpha, void* ws, size_t ws_bytes, void* stream);
int vnet_bn_act_bwd_apply_b16(const void* dy16, const void* x, const void* r16, int bcast, int64_t M, int C, const float* mean,
    const float* invstd, const float* gamma, const float* beta, int act, const float* alpha, const float* sum_dz,
    const float* sum_dz_xhat, double M_total, const float* xhat_coef, void* ds16, void* stream);
/* Small tensors (M <= 8192 rows: the 16^3 / 8^3 levels): the whole batch-norm of a layer in ONE launch per direction (one workgroup
 * per channel octet; replaces statistics rows + finalize + vnet_bn_act_fwd_b16, and reduce + finalize + apply).  Per-replica only.
 * fwd: mean / invstd written (+ moving averages), y = act(BN(x + r)).  bwd: dgamma / dbeta / dalpha, ds16 (NULL to skip). */
int vnet_bn_small_ok(int64_t M, int C);
int vnet_bn_small_fwd_b16(const void* x16, const void* r16, int64_t M, int C, float eps, float momentum, const float* gamma,
    const float* beta, int act, const float* alpha, float* mean, float* invstd, float* moving_mean, float* moving_var, void* y16,
    void* stream);
int vnet_bn_small_bwd_b16(const void* dy16, const void* x16, const void* r16, int64_t M, int C, const float* mean,
    const float* invstd, const float* gamma, const float* beta, int act, const float* alpha, float* dgamma, float* dbeta,
    float* dalpha, void* ds16, void* stream);
/* 1x1x1 output head: bf16 activations in, fp32 logits out (K <= 8); backward: fp32 dy, bf16 dx (NULL to skip), fp32 dw / db */
int vnet_head_fwd_b16(const void* x16, const float* w, const float* bias, float* y, int64_t M, int C, int K, void* stream);
int vnet_head_bwd_b16(const void* x16, const float* w, const float* dy, void* dx16, float* dw, float* db, int64_t M, int C, int K,
    void* ws, size_t ws_bytes, void* stream);
/* dropout on bf16 tensors (n % 8 == 0); the mask stream is the fp32 kernel's (same seed -> same mask) */
int vnet_dropout_fwd_b16(const void* x16, void* y16, uint8_t* mask, int64_t n, float rate, uint64_t seed, const void* state,
    void* stream);
int vnet_dropout_bwd_b16(const void* dy16, const uint8_t* mask, void* dx16, int64_t n, float rate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VNET_HIP_H */

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

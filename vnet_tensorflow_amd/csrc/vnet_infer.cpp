// vnet_infer.cpp -- native sliding-window inference driver on top of the C ABI (include/vnet_hip.h).
//
// The MI355X-native counterpart of the reference's deprecated cxx/ demo (tf_inference.cpp:96-476, ThreadPool.h,
// SafeQueue.h: TF-1.8 C++ API + ITK): load a weights blob, enumerate overlapping patches exactly like
// evaluate_single_3D (model.py:866-903, including the duplicated last batch), crop patches on a CPU thread pool
// into pinned buffers, double-buffer the H2D copies against the forward pass, run networks.VNet (networks.py:246-365,
// batch statistics like the reference's train_phase=True at model.py:917), accumulate the softmax and the hit count on
// the GPU (model.py:919-929), take argmax of the SUMMED softmax (model.py:934), optionally normalise the
// probabilities (model.py:935-937), write .npy volumes.  No TF, no ITK: volumes are float32 .npy [X,Y,Z] or [X,Y,Z,C].
//
//   vnet_infer --weights net.vnetw --image vol.npy --label-out label.npy [--prob-out prob.npy]
//              --classes 2 --channels 16 --levels 4 --convs 1,2,3,3 --bottom 3 --patch 64,64,64 --stride 32,32,32 --batch 2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <queue>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vnet_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)
#define ABI_OK(x) do { int e_ = (x); if (e_ != 0) { std::fprintf(stderr, "libvnet_hip error %d at %s:%d (%s)\n", e_, __FILE__, __LINE__, #x); std::exit(3); } } while (0)

// ---- thread pool (cropping workers; the reference uses hardware_concurrency() workers too, tf_inference.cpp:363) ----
class ThreadPool {
public:
    explicit ThreadPool(size_t n) {
        for (size_t i = 0; i < n; ++i)
            workers_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                        if (stop_ && jobs_.empty()) return;
                        job = std::move(jobs_.front());
                        jobs_.pop();
                    }
                    job();
                }
            });
    }
    std::future<void> submit(std::function<void()> f) {
        auto task = std::make_shared<std::packaged_task<void()>>(std::move(f));
        std::future<void> fut = task->get_future();
        { std::lock_guard<std::mutex> lk(m_); jobs_.emplace([task] { (*task)(); }); }
        cv_.notify_one();
        return fut;
    }
    ~ThreadPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
private:
    std::vector<std::thread> workers_;
    std::queue<std::function<void()>> jobs_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

// ---- minimal .npy (v1/v2, little-endian, C order) -----------------------------------------------------------------
struct Npy { std::vector<int64_t> shape; std::vector<float> data; };

static Npy read_npy_f32(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(1); }
    char magic[6]; f.read(magic, 6);
    if (std::memcmp(magic, "\x93NUMPY", 6) != 0) { std::fprintf(stderr, "%s: not a .npy file\n", path.c_str()); std::exit(1); }
    unsigned char ver[2]; f.read((char*)ver, 2);
    uint32_t hlen = 0;
    if (ver[0] == 1) { uint16_t h; f.read((char*)&h, 2); hlen = h; } else { f.read((char*)&hlen, 4); }
    std::string hdr(hlen, ' '); f.read(&hdr[0], hlen);
    if (hdr.find("'<f4'") == std::string::npos || hdr.find("'fortran_order': False") == std::string::npos) {
        std::fprintf(stderr, "%s: need little-endian float32, C order (got %s)\n", path.c_str(), hdr.c_str()); std::exit(1);
    }
    Npy out;
    size_t a = hdr.find('(', hdr.find("'shape'")), b = hdr.find(')', a);
    std::stringstream ss(hdr.substr(a + 1, b - a - 1));
    std::string tok;
    while (std::getline(ss, tok, ',')) { if (tok.find_first_of("0123456789") != std::string::npos) out.shape.push_back(std::stoll(tok)); }
    size_t n = 1; for (auto d : out.shape) n *= (size_t)d;
    out.data.resize(n);
    f.read((char*)out.data.data(), n * 4);
    if ((size_t)f.gcount() != n * 4) { std::fprintf(stderr, "%s: truncated\n", path.c_str()); std::exit(1); }
    return out;
}

static void write_npy(const std::string& path, const char* descr, const std::vector<int64_t>& shape, const void* data, size_t bytes) {
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); ++i) sh += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    f.write("\x93NUMPY\x01\x00", 8);
    uint16_t h = (uint16_t)hdr.size(); f.write((char*)&h, 2);
    f.write(hdr.data(), hdr.size());
    f.write((const char*)data, bytes);
}

// ---- weights blob: "VNETW1\0\0", u32 nvars, then {u32 name_len, name, u32 ndim, u32 dims[], f32 data} ----------------
struct Var { std::vector<uint32_t> dims; float* dev = nullptr; size_t n = 0; };

static std::map<std::string, Var> load_weights(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(1); }
    char magic[8]; f.read(magic, 8);
    if (std::memcmp(magic, "VNETW1\0\0", 8) != 0) { std::fprintf(stderr, "%s: bad magic\n", path.c_str()); std::exit(1); }
    uint32_t nv; f.read((char*)&nv, 4);
    std::map<std::string, Var> vars;
    for (uint32_t i = 0; i < nv; ++i) {
        uint32_t nl; f.read((char*)&nl, 4);
        std::string name(nl, ' '); f.read(&name[0], nl);
        uint32_t nd; f.read((char*)&nd, 4);
        Var v; v.dims.resize(nd); f.read((char*)v.dims.data(), nd * 4);
        v.n = 1; for (auto d : v.dims) v.n *= d;
        std::vector<float> host(v.n); f.read((char*)host.data(), v.n * 4);
        HIP_OK(hipMalloc((void**)&v.dev, std::max<size_t>(v.n, 4) * 4));
        HIP_OK(hipMemcpy(v.dev, host.data(), v.n * 4, hipMemcpyHostToDevice));
        vars[name] = v;
    }
    return vars;
}

// ---- the network (forward only), wired on the C ABI ----------------------------------------------------------------
struct Tensor {
    float* p;        // fp32 data (null for a bf16-storage tensor)
    int B, D, H, W, C;
    void* q = nullptr;   // bf16 data (--compute bf16: bf16 tensors end to end, vnet_hip.h *_b16)
    size_t numel() const { return (size_t)B * D * H * W * C; }
    int64_t rows() const { return (int64_t)B * D * H * W; }
};

struct Config {
    int classes = 2, channels = 16, levels = 4, bottom = 3, batch = 1;
    std::vector<int> convs{1, 2, 3, 3};
    int patch[3] = {64, 64, 64}, stride[3] = {64, 64, 64};
    std::string weights, image, label_out, prob_out;
    bool normalise = true;
    bool split3 = false;     // --compute fp32_split3: fp32 tensors, the 5^3 convolutions on the bf16 matrix pipe (vnet_conv_fwd_x3) where vnet_conv_x3_ok
    bool store16 = false;    // --compute bf16: every activation is a bf16 tensor (BASELINE config C5 as the Python path runs it)
};

class VNetForward {
public:
    VNetForward(const Config& c, std::map<std::string, Var>& vars, hipStream_t st) : cfg(c), vars_(vars), st_(st) {
        HIP_OK(hipMalloc(&ws_, ws_bytes_));
        HIP_OK(hipMalloc((void**)&stat_, 4096 * sizeof(float)));
    }
    void set_arena(size_t bytes) { HIP_OK(hipMalloc((void**)&arena_, bytes)); arena_bytes_ = bytes; }

    // images [B,P0,P1,P2,Cin] on the device -> softmax [B,P0,P1,P2,K] (arena memory, valid until the next call)
    Tensor forward(const Tensor& images) {
        top_ = 0; scope_.clear(); bn_count_.clear();
        Tensor x = images;
        scope_ = {"vnet/input_layer"};
        if (images.C == 1) x = bn(x, 0, nullptr, true);
        else {
            if (cfg.store16) x = cast16(x);       // bf16, channels zero-padded to the 16-byte unit (ops.cast_input)
            x = conv(x, nullptr, 5, 1, cfg.channels, images.C);
            x = bn(x, VNET_ACT_PRELU, nullptr, false);
        }
        std::vector<Tensor> feats;
        for (int l = 0; l < cfg.levels; ++l) {
            scope_ = {"vnet/encoder/level_" + std::to_string(l + 1)};
            x = block(x, cfg.convs[l]);
            feats.push_back(x);
            scope_.push_back("down_convolution");
            x = conv(x, nullptr, 2, 2, x.C * 2);
            x = bn(x, VNET_ACT_PRELU, nullptr, false);
        }
        scope_ = {"vnet/bottom_level"};
        x = block(x, cfg.bottom);
        for (int l = cfg.levels - 1; l >= 0; --l) {
            scope_ = {"vnet/decoder/level_" + std::to_string(l + 1), "up_convolution"};
            x = upconv(x, feats[l]);
            x = bn(x, VNET_ACT_PRELU, nullptr, false);
            scope_.pop_back();
            x = block2(x, feats[l], cfg.convs[l]);
        }
        scope_ = {"vnet/output_layer"};
        Tensor logits = head(x);
        logits = bn(logits, VNET_ACT_NONE, nullptr, false);
        return softmax(logits);
    }

private:
    const Config& cfg;
    std::map<std::string, Var>& vars_;
    hipStream_t st_;
    std::vector<std::string> scope_;
    std::map<std::string, int> bn_count_;
    std::map<std::string, float*> packed_;
    char* arena_ = nullptr; size_t arena_bytes_ = 0, top_ = 0;
    void* ws_ = nullptr; size_t ws_bytes_ = (size_t)768 << 20;
    float* stat_ = nullptr;

    std::string scope() const { std::string s; for (auto& p : scope_) s += (s.empty() ? "" : "/") + p; return s; }
    Var& var(const std::string& name) {
        auto it = vars_.find(name);
        if (it == vars_.end()) { std::fprintf(stderr, "weights blob has no variable %s\n", name.c_str()); std::exit(1); }
        return it->second;
    }
    Tensor alloc(int B, int D, int H, int W, int C) {
        Tensor t{nullptr, B, D, H, W, C};
        size_t bytes = (t.numel() * 4 + 255) / 256 * 256;
        if (top_ + bytes > arena_bytes_) { std::fprintf(stderr, "activation arena too small\n"); std::exit(1); }
        t.p = (float*)(arena_ + top_); top_ += bytes;
        return t;
    }
    Tensor alloc16(int B, int D, int H, int W, int C) {
        Tensor t{nullptr, B, D, H, W, C};
        size_t bytes = (t.numel() * 2 + 255) / 256 * 256;
        if (top_ + bytes > arena_bytes_) { std::fprintf(stderr, "activation arena too small\n"); std::exit(1); }
        t.q = arena_ + top_; top_ += bytes;
        return t;
    }
    Tensor cast16(const Tensor& x) {
        const int Cp = (x.C + 7) / 8 * 8;
        Tensor y = alloc16(x.B, x.D, x.H, x.W, Cp);
        ABI_OK(vnet_cast_bf16(x.p, y.q, x.rows(), x.C, Cp, st_));
        return y;
    }
    float* pack(const std::string& wname, int mode, int taps, int I, int O) {
        const std::string key = wname + "#" + std::to_string(mode);
        auto it = packed_.find(key);
        if (it != packed_.end()) return it->second;
        float* wp; HIP_OK(hipMalloc((void**)&wp, vnet_packed_weight_floats(mode, taps, I, O) * 4));
        ABI_OK(vnet_pack_weights(mode, var(wname).dev, wp, taps, I, O, st_));
        return packed_[key] = wp;          // inference: filters never change, pack once
    }
    // tf.layers.batch_normalization(training=True) [+ residual] [+ tile] + activation; alpha lives in the enclosing scope
    Tensor bn(const Tensor& x, int act, const Tensor* res, bool tile) {
        const std::string sc = scope();
        int n = bn_count_[sc]++;
        const std::string name = sc + "/batch_normalization" + (n ? "_" + std::to_string(n) : "");
        Var& g = var(name + "/gamma"); Var& b = var(name + "/beta");
        const int C = (int)g.n;
        float* mean = stat_; float* invstd = stat_ + 1024;
        const float* alpha = act == VNET_ACT_PRELU ? var(sc + "/alpha").dev : nullptr;
        if (cfg.store16 && (x.q || tile)) {
            // bf16 storage: statistics of the bf16 tensor (+ bf16 residual) -- or of the fp32 1-channel image that is tiled -- in fp32,
            // one rounding of the normalised / activated value
            Tensor y = alloc16(x.B, x.D, x.H, x.W, C);
            // tiny tensors: one launch -- the Python path's rule (ops._SMALL_BN: on, <= 512 rows)
            constexpr bool small_on = true;
            constexpr long small_rows = 512;
            if (!tile && small_on && (long)x.rows() <= small_rows && vnet_bn_small_ok(x.rows(), C)) {
                ABI_OK(vnet_bn_small_fwd_b16(x.q, res ? res->q : nullptr, x.rows(), C, 1e-3f, 0.99f, g.dev, b.dev, act, alpha, mean, invstd,
                                             nullptr, nullptr, y.q, st_));
                return y;
            }
            if (tile) ABI_OK(vnet_bn_stats(x.p, nullptr, 1, x.rows(), C, 1e-3f, 0.99f, mean, invstd, nullptr, nullptr, ws_, ws_bytes_, st_));
            else ABI_OK(vnet_bn_stats_b16(x.q, res ? res->q : nullptr, x.rows(), C, 1e-3f, 0.99f, mean, invstd, nullptr, nullptr, ws_, ws_bytes_, st_));
            ABI_OK(vnet_bn_act_fwd_b16(tile ? (const void*)x.p : x.q, res ? res->q : nullptr, tile ? 1 : 0, x.rows(), C, mean, invstd, g.dev, b.dev,
                                       act, alpha, y.q, st_));
            return y;
        }
        Tensor y = alloc(x.B, x.D, x.H, x.W, C);
        ABI_OK(vnet_bn_stats(x.p, res ? res->p : nullptr, tile ? 1 : 0, x.rows(), C, 1e-3f, 0.99f, mean, invstd, nullptr, nullptr, ws_, ws_bytes_, st_));
        ABI_OK(vnet_bn_act_fwd(x.p, res ? res->p : nullptr, tile ? 1 : 0, x.rows(), C, mean, invstd, g.dev, b.dev, act, alpha, y.p, st_));
        return y;
    }
    // the decoder's batch-norm chains in closed form (include/vnet_hip.h, vnet_bn_chain_coef_fwd): one fused normalisation of x
    //   kind 0: x = BN(x); r = BN(x); out = prelu(BN(x + r))      kind 1: r = BN(x); out = prelu(BN(x + r))
    Tensor bn_chain(const Tensor& x, int kind) {
        const std::string sc = scope();
        const float* gp[3] = {nullptr, nullptr, nullptr};
        const float* bp[3] = {nullptr, nullptr, nullptr};
        int C = 0;
        for (int k = 0; k < (kind == 0 ? 3 : 2); ++k) {
            int n = bn_count_[sc]++;
            const std::string name = sc + "/batch_normalization" + (n ? "_" + std::to_string(n) : "");
            Var& g = var(name + "/gamma"); Var& b = var(name + "/beta");
            gp[k] = g.dev; bp[k] = b.dev; C = (int)g.n;
        }
        float* mean = stat_; float* invstd = stat_ + 1024; float* ceff = stat_ + 2048; float* deff = stat_ + 3072;
        if (x.q) {
            Tensor y = alloc16(x.B, x.D, x.H, x.W, C);
            ABI_OK(vnet_bn_stats_b16(x.q, nullptr, x.rows(), C, 1e-3f, 0.99f, mean, invstd, nullptr, nullptr, ws_, ws_bytes_, st_));
            ABI_OK(vnet_bn_chain_coef_fwd(kind, C, 1e-3f, 0.99f, mean, invstd, gp[0], bp[0], gp[1], bp[1], gp[2], bp[2], ceff, deff,
                                          nullptr, nullptr, nullptr, nullptr, st_));
            ABI_OK(vnet_bn_act_fwd_b16(x.q, nullptr, 0, x.rows(), C, mean, invstd, ceff, deff, VNET_ACT_PRELU, var(sc + "/alpha").dev, y.q, st_));
            return y;
        }
        Tensor y = alloc(x.B, x.D, x.H, x.W, C);
        ABI_OK(vnet_bn_stats(x.p, nullptr, 0, x.rows(), C, 1e-3f, 0.99f, mean, invstd, nullptr, nullptr, ws_, ws_bytes_, st_));
        ABI_OK(vnet_bn_chain_coef_fwd(kind, C, 1e-3f, 0.99f, mean, invstd, gp[0], bp[0], gp[1], bp[1], gp[2], bp[2], ceff, deff,
                                      nullptr, nullptr, nullptr, nullptr, st_));
        ABI_OK(vnet_bn_act_fwd(x.p, nullptr, 0, x.rows(), C, mean, invstd, ceff, deff, VNET_ACT_PRELU, var(sc + "/alpha").dev, y.p, st_));
        return y;
    }
    // Cin_w: input channels of the FILTER when the tensor carries zero-padded channels (the cast 4-modality input), else 0
    Tensor conv(const Tensor& x0, const Tensor* x1, int ks, int stride, int Cout, int Cin_w = 0) {
        const std::string sc = scope();
        const int Cin = x0.C + (x1 ? x1->C : 0);
        const int Do = (x0.D + stride - 1) / stride, Ho = (x0.H + stride - 1) / stride, Wo = (x0.W + stride - 1) / stride;
        if (x0.q) {
            Tensor y = alloc16(x0.B, Do, Ho, Wo, Cout);
            if (ks == 5 && stride == 1) {
                float* wpb = pack(sc + "/weights", VNET_PACK_FWD_BF16, 125, Cin_w ? Cin_w : Cin, Cout);
                if (vnet_conv_b16_ws_bytes(x0.C, x1 ? x1->C : 0, Cout, 0, x0.B, Do, Ho, Wo) > ws_bytes_) { std::fprintf(stderr, "workspace too small\n"); std::exit(1); }
                if (Cin_w && !x1)       // the zero-padded network input: x-im2col form where the shape allows (as ops._conv5_b16_call)
                    ABI_OK(vnet_conv_fwd_b16_padded(x0.q, x0.C, Cin_w, wpb, var(sc + "/biases").dev, y.q, Cout, x0.B, x0.D, x0.H, x0.W,
                                                    nullptr, nullptr, ws_, ws_bytes_, st_));
                else
                    ABI_OK(vnet_conv_fwd_b16(x0.q, x0.C, x1 ? x1->q : nullptr, x1 ? x1->C : 0, wpb, var(sc + "/biases").dev, y.q, Cout, nullptr, 0,
                                             x0.B, x0.D, x0.H, x0.W, nullptr, nullptr, nullptr, ws_, ws_bytes_, st_));
            } else if (vnet_conv2_direct_ok(Cin, Cout)) {     // 2^3 stride-2 at levels 1-2: LDS-free kernel on the unpacked filter
                ABI_OK(vnet_conv2_direct_b16(1, x0.q, y.q, var(sc + "/weights").dev, var(sc + "/biases").dev, Cin, Cout, x0.B, x0.D, x0.H, x0.W,
                                             Do, Ho, Wo, 0, nullptr, st_));
            } else {
                float* wp = pack(sc + "/weights", VNET_PACK_FWD | VNET_PACK_ROUND_BF16, 8, Cin, Cout);
                ABI_OK(vnet_conv2_fwd_b16(0, x0.q, Cin, wp, var(sc + "/biases").dev, y.q, Cout, x0.B, x0.D, x0.H, x0.W, Do, Ho, Wo, 0, nullptr,
                                          ws_, ws_bytes_, st_));
            }
            return y;
        }
        Tensor y = alloc(x0.B, Do, Ho, Wo, Cout);
        if (cfg.split3 && ks == 5 && stride == 1 && vnet_conv_x3_ok(x0.C, x1 ? x1->C : 0, Cout, 0, x0.B, Do, Ho, Wo) == 1) {
            // the Python path's rule (ops._x3_ok): exactly split bf16 operands, six products, fp32 accumulate
            float* wp3 = pack(sc + "/weights", VNET_PACK_FWD_X3, 125, Cin, Cout);
            if (vnet_conv_x3_ws_bytes(Cin, Cout, x0.B, Do, Ho, Wo) > ws_bytes_) { std::fprintf(stderr, "workspace too small\n"); std::exit(1); }
            ABI_OK(vnet_conv_fwd_x3(x0.p, x0.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, wp3, var(sc + "/biases").dev, y.p, Cout, nullptr, 0,
                                    x0.B, x0.D, x0.H, x0.W, nullptr, nullptr, nullptr, ws_, ws_bytes_, st_));
            return y;
        }
        float* wp = pack(sc + "/weights", VNET_PACK_FWD, ks * ks * ks, Cin, Cout);
        size_t need = vnet_conv_ws_bytes(ks, 0, stride, 0, Cin, Cout, x0.B, Do, Ho, Wo);
        if (need > ws_bytes_) { std::fprintf(stderr, "workspace too small\n"); std::exit(1); }
        ABI_OK(vnet_conv_fwd(ks, 0, stride, 0, x0.p, x0.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, wp, var(sc + "/biases").dev,
                             y.p, Cout, nullptr, 0, x0.B, x0.D, x0.H, x0.W, Do, Ho, Wo, ws_, ws_bytes_, st_));
        return y;
    }
    Tensor upconv(const Tensor& x, const Tensor& like) {
        const std::string sc = scope();
        const int Cout = x.C / 2;
        if (x.q) {
            Tensor y = alloc16(x.B, like.D, like.H, like.W, Cout);
            if (vnet_conv2_direct_ok(Cout, x.C)) {
                ABI_OK(vnet_conv2_direct_b16(0, x.q, y.q, var(sc + "/weights").dev, var(sc + "/biases").dev, Cout, x.C, x.B, like.D, like.H, like.W,
                                             x.D, x.H, x.W, 0, nullptr, st_));
            } else {
                float* wp = pack(sc + "/weights", VNET_PACK_UP | VNET_PACK_ROUND_BF16, 8, x.C, Cout);
                ABI_OK(vnet_conv2_fwd_b16(1, x.q, x.C, wp, var(sc + "/biases").dev, y.q, Cout, x.B, x.D, x.H, x.W, like.D, like.H, like.W, 0, nullptr,
                                          ws_, ws_bytes_, st_));
            }
            return y;
        }
        Tensor y = alloc(x.B, like.D, like.H, like.W, Cout);
        float* wp = pack(sc + "/weights", VNET_PACK_UP, 8, x.C, Cout);
        ABI_OK(vnet_conv_fwd(2, 0, 2, 1, x.p, x.C, nullptr, 0, wp, var(sc + "/biases").dev, y.p, Cout, nullptr, 0,
                             x.B, x.D, x.H, x.W, like.D, like.H, like.W, ws_, ws_bytes_, st_));
        return y;
    }
    Tensor head(const Tensor& x) {
        const std::string sc = scope();
        Tensor y = alloc(x.B, x.D, x.H, x.W, cfg.classes);
        if (x.q) ABI_OK(vnet_head_fwd_b16(x.q, var(sc + "/weights").dev, var(sc + "/biases").dev, y.p, x.rows(), x.C, cfg.classes, st_));
        else ABI_OK(vnet_head_fwd(x.p, var(sc + "/weights").dev, var(sc + "/biases").dev, y.p, x.rows(), x.C, cfg.classes, st_));
        return y;
    }
    Tensor softmax(const Tensor& logits) {
        Tensor sm = alloc(logits.B, logits.D, logits.H, logits.W, logits.C);
        const int64_t V = (int64_t)logits.D * logits.H * logits.W;
        Tensor lab = alloc(logits.B, logits.D, logits.H, logits.W, 1);          // dummy labels (zeros) for the fused head
        HIP_OK(hipMemsetAsync(lab.p, 0, lab.numel() * 4, st_));
        float* scal = stat_ + 2048;
        ABI_OK(vnet_softmax_dice_fwd(logits.p, (const int32_t*)lab.p, logits.B, V, logits.C, VNET_LOSS_SORENSEN, nullptr, 1.f, 1e-5f,
                                     sm.p, nullptr, scal, scal + 1, scal + 8, ws_, ws_bytes_, st_));
        return sm;
    }
    // networks.py:307-322
    Tensor block(Tensor x, int n) {
        const Tensor input = x;
        for (int i = 0; i < n; ++i) {
            scope_.push_back("conv_" + std::to_string(i + 1));
            x = conv(x, nullptr, 5, 1, x.C);
            x = bn(x, VNET_ACT_PRELU, i == n - 1 ? &input : nullptr, false);
            scope_.pop_back();
        }
        return x;
    }
    // networks.py:324-365 (concat never materialised; the dead BN of non-last convs only updates moving stats -> skipped)
    Tensor block2(const Tensor& up, const Tensor& skip, int n) {
        const int C = up.C;
        scope_.push_back("conv_1");
        Tensor x = conv(up, &skip, 5, 1, C);
        if (n == 1) {
            x = bn_chain(x, 0);
            scope_.pop_back();
            return x;
        }
        x = bn(x, VNET_ACT_PRELU, nullptr, false);
        scope_.pop_back();
        for (int i = 1; i < n; ++i) {
            scope_.push_back("conv_" + std::to_string(i + 1));
            x = conv(x, nullptr, 5, 1, C);
            if (i == n - 1) {
                x = bn_chain(x, 1);
            } else {
                bn_count_[scope()]++;                       // the unused residual-branch BN still owns the first layer name
                x = bn(x, VNET_ACT_PRELU, nullptr, false);
            }
            scope_.pop_back();
        }
        return x;
    }
};

// ---- CLI ---------------------------------------------------------------------------------------------------------------
static std::vector<int> ints(const std::string& s) { std::vector<int> v; std::stringstream ss(s); std::string t; while (std::getline(ss, t, ',')) v.push_back(std::stoi(t)); return v; }

static Config parse(int argc, char** argv) {
    Config c;
    for (int i = 1; i + 1 < argc + 1; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); } return argv[++i]; };
        if (a == "--weights") c.weights = next(); else if (a == "--image") c.image = next();
        else if (a == "--label-out") c.label_out = next(); else if (a == "--prob-out") c.prob_out = next();
        else if (a == "--classes") c.classes = std::stoi(next()); else if (a == "--channels") c.channels = std::stoi(next());
        else if (a == "--levels") c.levels = std::stoi(next()); else if (a == "--bottom") c.bottom = std::stoi(next());
        else if (a == "--batch") c.batch = std::stoi(next()); else if (a == "--convs") c.convs = ints(next());
        else if (a == "--patch") { auto v = ints(next()); for (int k = 0; k < 3; ++k) c.patch[k] = v[k]; }
        else if (a == "--stride") { auto v = ints(next()); for (int k = 0; k < 3; ++k) c.stride[k] = v[k]; }
        else if (a == "--no-normalise") c.normalise = false;
        else if (a == "--compute") {
            const std::string v = next();
            if (v != "fp32" && v != "fp32_split3" && v != "bf16") { std::fprintf(stderr, "--compute fp32|fp32_split3|bf16\n"); std::exit(1); }
            c.store16 = (v == "bf16"); c.split3 = (v == "fp32_split3");
        }
        else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); std::exit(1); }
    }
    if (c.store16 && (c.channels < 8 || (c.channels & (c.channels - 1)))) {
        std::fprintf(stderr, "--compute bf16 needs --channels 8 * 2^k (16-byte units of bf16 channels); use fp32\n");
        std::exit(1);
    }
    if (c.weights.empty() || c.image.empty() || c.label_out.empty() || (int)c.convs.size() != c.levels) {
        std::fprintf(stderr, "usage: vnet_infer --weights W --image I.npy --label-out L.npy [--prob-out P.npy] --classes K --channels C "
                             "--levels L --convs a,b,.. --bottom n --patch x,y,z --stride x,y,z --batch b [--compute fp32|fp32_split3|bf16]\n");
        std::exit(1);
    }
    return c;
}

int main(int argc, char** argv) {
    Config cfg = parse(argc, argv);
    Npy img = read_npy_f32(cfg.image);
    if (img.shape.size() == 3) img.shape.push_back(1);
    if (img.shape.size() != 4) { std::fprintf(stderr, "image must be [X,Y,Z] or [X,Y,Z,C]\n"); return 1; }
    const int X = (int)img.shape[0], Y = (int)img.shape[1], Z = (int)img.shape[2], Cin = (int)img.shape[3], K = cfg.classes;
    const int P0 = cfg.patch[0], P1 = cfg.patch[1], P2 = cfg.patch[2];
    if (X < P0 || Y < P1 || Z < P2) { std::fprintf(stderr, "volume smaller than the patch (pad it first)\n"); return 1; }

    hipStream_t compute, copy;
    HIP_OK(hipStreamCreate(&compute)); HIP_OK(hipStreamCreate(&copy));
    auto vars = load_weights(cfg.weights);
    VNetForward net(cfg, vars, compute);
    const size_t patch_vox = (size_t)P0 * P1 * P2;
    net.set_arena((size_t)cfg.batch * patch_vox * 4 * (size_t)(cfg.channels * 14 + 64) + ((size_t)64 << 20));

    // patch enumeration, model.py:866-903 (last patch clamped to the border; the last batch is appended twice)
    int num[3]; const int dims[3] = {X, Y, Z};
    for (int a = 0; a < 3; ++a) num[a] = (int)std::ceil((dims[a] - cfg.patch[a]) / (double)cfg.stride[a]) + 1;
    std::vector<std::vector<std::array<int, 3>>> batches;
    {
        std::vector<std::array<int, 3>> cur; int total = 0;
        std::vector<size_t> open_idx;
        for (int i = 0; i < num[0]; ++i) for (int j = 0; j < num[1]; ++j) for (int k = 0; k < num[2]; ++k) {
            if (total % cfg.batch == 0) { batches.emplace_back(); }
            std::array<int, 3> s{i * cfg.stride[0], j * cfg.stride[1], k * cfg.stride[2]};
            for (int a = 0; a < 3; ++a) if (s[a] + cfg.patch[a] > dims[a]) s[a] = dims[a] - cfg.patch[a];
            batches.back().push_back(s);
            ++total;
        }
        batches.push_back(batches.back());
    }

    float *d_vol, *d_cnt;
    const size_t nvox = (size_t)X * Y * Z;
    HIP_OK(hipMalloc((void**)&d_vol, nvox * K * 4)); HIP_OK(hipMemset(d_vol, 0, nvox * K * 4));
    HIP_OK(hipMalloc((void**)&d_cnt, nvox * 4)); HIP_OK(hipMemset(d_cnt, 0, nvox * 4));

    // two pinned staging buffers + two device input buffers: crop(i+1) and H2D(i+1) overlap forward(i)
    const size_t batch_floats = (size_t)cfg.batch * patch_vox * Cin;
    float* h_in[2]; float* d_in[2]; hipEvent_t copied[2], consumed[2];
    for (int s = 0; s < 2; ++s) {
        HIP_OK(hipHostMalloc((void**)&h_in[s], batch_floats * 4, hipHostMallocDefault));
        HIP_OK(hipMalloc((void**)&d_in[s], batch_floats * 4));
        HIP_OK(hipEventCreate(&copied[s])); HIP_OK(hipEventCreate(&consumed[s]));
    }
    ThreadPool pool(std::max(2u, std::thread::hardware_concurrency() / 2));
    auto crop = [&](size_t bi, int slot) {
        std::vector<std::future<void>> futs;
        for (size_t p = 0; p < batches[bi].size(); ++p)
            futs.push_back(pool.submit([&, p, bi, slot] {
                const auto s = batches[bi][p];
                float* dst = h_in[slot] + p * patch_vox * Cin;
                for (int x = 0; x < P0; ++x) for (int y = 0; y < P1; ++y) {
                    const float* src = img.data.data() + (((size_t)(s[0] + x) * Y + (s[1] + y)) * Z + s[2]) * Cin;
                    std::memcpy(dst + ((size_t)x * P1 + y) * P2 * Cin, src, (size_t)P2 * Cin * 4);
                }
            }));
        for (auto& f : futs) f.get();
    };
    auto upload = [&](size_t bi, int slot) {
        HIP_OK(hipStreamWaitEvent(copy, consumed[slot], 0));
        HIP_OK(hipMemcpyAsync(d_in[slot], h_in[slot], batches[bi].size() * patch_vox * Cin * 4, hipMemcpyHostToDevice, copy));
        HIP_OK(hipEventRecord(copied[slot], copy));
    };
    for (int s = 0; s < 2; ++s) HIP_OK(hipEventRecord(consumed[s], compute));
    const auto t_start = std::chrono::steady_clock::now();
    crop(0, 0); upload(0, 0);
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        const int slot = (int)(bi & 1);
        std::future<void> next;
        if (bi + 1 < batches.size())
            next = std::async(std::launch::async, [&, bi] { HIP_OK(hipEventSynchronize(consumed[(bi + 1) & 1])); crop(bi + 1, (int)((bi + 1) & 1)); });
        HIP_OK(hipStreamWaitEvent(compute, copied[slot], 0));
        Tensor in{d_in[slot], (int)batches[bi].size(), P0, P1, P2, Cin};
        Tensor sm = net.forward(in);
        for (size_t p = 0; p < batches[bi].size(); ++p)
            ABI_OK(vnet_accumulate_patch(sm.p + p * patch_vox * K, d_vol, d_cnt, K, P0, P1, P2, batches[bi][p][0], batches[bi][p][1],
                                         batches[bi][p][2], X, Y, Z, compute));
        HIP_OK(hipEventRecord(consumed[slot], compute));
        if (next.valid()) { next.get(); upload(bi + 1, (int)((bi + 1) & 1)); }
    }
    HIP_OK(hipStreamSynchronize(compute));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    size_t npatch = 0;
    for (auto& bt : batches) npatch += bt.size();

    std::vector<float> vol(nvox * K), cnt(nvox);
    HIP_OK(hipMemcpy(vol.data(), d_vol, nvox * K * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(cnt.data(), d_cnt, nvox * 4, hipMemcpyDeviceToHost));
    std::vector<int16_t> label(nvox);
    for (size_t v = 0; v < nvox; ++v) {                   // argmax of the summed softmax, first maximum on ties (np.argmax)
        int best = 0;
        for (int k = 1; k < K; ++k) if (vol[v * K + k] > vol[v * K + best]) best = k;
        label[v] = (int16_t)best;
    }
    write_npy(cfg.label_out, "<i2", {X, Y, Z}, label.data(), nvox * 2);
    if (!cfg.prob_out.empty()) {
        std::vector<float> prob(nvox * K);                // [K,X,Y,Z] like the reference's per-class probability volumes
        for (size_t v = 0; v < nvox; ++v) for (int k = 0; k < K; ++k) prob[(size_t)k * nvox + v] = cfg.normalise ? vol[v * K + k] / cnt[v] : vol[v * K + k];
        write_npy(cfg.prob_out, "<f4", {K, X, Y, Z}, prob.data(), nvox * K * 4);
    }
    std::printf("vnet_infer: %zu batches (%zu patches of %dx%dx%d), %dx%dx%d volume, %d classes -> %s\n", batches.size(), npatch,
                P0, P1, P2, X, Y, Z, K, cfg.label_out.c_str());
    std::printf("vnet_infer: sliding window %.3f s = %.1f patches/s (crop + H2D + forward + accumulate, %s)\n", secs, npatch / secs,
                cfg.store16 ? "bf16 storage" : cfg.split3 ? "fp32_split3" : "fp32");
    return 0;
}

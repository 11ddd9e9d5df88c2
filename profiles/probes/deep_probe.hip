// Stand-alone harness of the deep-level bf16 convolution kernel (vnet_tensorflow_amd/csrc/conv_deep.h): times one layer shape
// (kernel + its split-K reduce, as the library launches them) and, built with -DDEEP_STAMPS, prints s_memtime stamps of the eight
// waves of one workgroup at the phase boundaries.  Ablations: -DDEEP_NO_A / -DDEEP_NO_B / -DDEEP_NO_MFMA (timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DDEEP_STAMPS] profiles/probes/deep_probe.hip -o profiles/probes/deep_probe
//   ./deep_probe P Cin Cout [iters]
#include "../../vnet_tensorflow_amd/csrc/conv_deep.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 32, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 64;
    const int iters = argc > 4 ? atoi(argv[4]) : 200;
    const size_t nvox = (size_t)P * P * P;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<unsigned short> hx(nvox * Cin), hw((size_t)125 * Cin * Cout);
    for (auto& v : hx) v = f2bf(U(rng));
    for (auto& v : hw) v = f2bf(U(rng) * 0.05f);
    unsigned short *x, *w, *y; float* ws; float* bias;
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&y, nvox * Cout * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));       // (any bits in fragment order: timing only)
    CK(hipMalloc(&bias, Cout * 4)); CK(hipMemset(bias, 0, Cout * 4));
    DeepPlan dp = plan_conv_deep(Cin, 0, Cout, 0, 1, P, P, P);
    if (!dp.use) { std::printf("deep plan not taken\n"); return 1; }
    const size_t wsb = (size_t)dp.nsplit * nvox * Cout * 4;
    CK(hipMalloc(&ws, wsb ? wsb : 16));
    ConvArgs a{};
    a.x0 = reinterpret_cast<const float*>(x); a.C0 = Cin; a.C1 = 0; a.Cin = Cin;
    a.wp = reinterpret_cast<const float4*>(w); a.bias = bias;
    a.y0 = reinterpret_cast<float*>(y); a.Cy0 = Cout; a.Cout = Cout;
    a.B = 1; a.Di = a.Hi = a.Wi = a.Do = a.Ho = a.Wo = P;
    a.nchunks = Cin / 16; a.CQ = a.nchunks * 4; a.CoutP = Cout; a.vec_in = a.vec_out = 1; a.pad = a.padx = 2;
    a.nbz = dp.nbz; a.nby = dp.nby; a.nbx = dp.nbx; a.cps = dp.cps; a.nz = 1;
    if (dp.nsplit > 1) { a.part = ws; a.part_stride = nvox * Cout; }
#ifdef DEEP_STAMPS
    long long* st; CK(hipMalloc(&st, 128 * 8)); CK(hipMemset(st, 0, 128 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_deep_stamps), &st, sizeof(st)));
#endif
    auto run = [&]() {
        if (int e = launch_conv_deep(a, dp, 0)) { std::fprintf(stderr, "launch %d\n", e); std::exit(1); }
        if (dp.nsplit > 1) {
            const size_t total = nvox * Cout;
            const int blocks = (int)std::min((size_t)2048, (total + 255) / 256);
            hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3(blocks), dim3(256), 0, 0, a.part, a.part_stride, dp.nsplit, bias,
                               y, (unsigned short*)nullptr, Cout, 0, a.CoutP, nvox, 0, (const unsigned short*)nullptr, (float*)nullptr,
                               (const unsigned short*)nullptr);
        }
    };
    for (int i = 0; i < 5; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / iters);
    }
    const double fl = 2.0 * nvox * 125 * Cin * Cout;
    std::printf("deep %d^3 %d->%d: grid %d x %d x %d, cps %d: %.2f us  %.1f TF/s\n", P, Cin, Cout, dp.nbz * dp.nby * dp.nbx, dp.ncob, dp.nsplit,
                dp.cps, best * 1e3, fl / best / 1e9);
#ifdef DEEP_STAMPS
    long long hs[128];
    CK(hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost));
    long long t0 = hs[0];
    for (int w8 = 0; w8 < 8; ++w8) t0 = std::min(t0, hs[w8 * 16]);
    std::printf("  cycles:   start | issue  commit barrier |  main  | wait  reduce | epilogue | end\n");
    for (int w8 = 0; w8 < 8; ++w8) {
        const long long* h = hs + w8 * 16;
        std::printf("  wave %d: %6lld | %6lld %6lld %6lld | %6lld | %6lld %6lld | %6lld | %6lld\n", w8, h[0] - t0, h[1] - h[0], h[2] - h[1], h[3] - h[2],
                    h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - t0);
    }
    std::printf("  one chunk: unit0  unit1  unit2  tail  commit barrier t_issue   (start at)\n");
    for (int w8 = 0; w8 < 8; ++w8) {
        const long long* h = hs + w8 * 16;
        std::printf("  wave %d: %6lld %6lld %6lld %6lld %6lld %6lld %6lld   (%lld)\n", w8, h[8] - (DEEP_STAMP_CHUNK ? h[8] : h[3]), h[9] - h[8], h[10] - h[9],
                    h[11] - h[10], h[12] - h[11], h[13] - h[12], h[14] - h[13], h[8] - t0);
    }
#endif
    return 0;
}

// Timing experiments on the split-operand network kernels (not part of the product): each kernel on
// synthetic data, timed with HIP events, with shader-clock stamps of one block's phases.
#define NNB_SPLIT_BENCH_CLOCKS 1
#include "../alphazero_amd/csrc/net_kernels_split.hip"
#include "conv_duo_experiment.hip"
#include "fc16_experiment.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace nnb;
namespace nnb {     // the heads stage lives in another translation unit of the library; never called here
void launch_net_stage_f16(hipStream_t, int, const NetWeights &, const NetBuffers &, const u64 *, const uint8_t *, float *,
                          float *, float *, const int32_t *, int) {}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> T *dev_random_halves(size_t n_halves, float lo, float hi, unsigned seed) {
    std::vector<_Float16> h(n_halves);
    unsigned s = seed * 2654435761u + 1;
    for (size_t i = 0; i < n_halves; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (_Float16)(lo + (hi - lo) * (float)(s >> 8) / 16777216.0f);
    }
    void *d; CK(hipMalloc(&d, n_halves * 2)); CK(hipMemcpy(d, h.data(), n_halves * 2, hipMemcpyHostToDevice));
    return (T *)d;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 3979, iters = 20;
    u32x4 *act2 = dev_random_halves<u32x4>((size_t)n * 72 * 256, 0.f, 1.f, 1);
    u32x4 *act3; CK(hipMalloc((void **)&act3, (size_t)n * 90 * 512));
    u32x4 *act2o; CK(hipMalloc((void **)&act2o, (size_t)n * 72 * 512));
    u32x4 *w2h = dev_random_halves<u32x4>(128 * 128 * 16, -1.f, 1.f, 2), *w2l = dev_random_halves<u32x4>(128 * 128 * 16, -1e-3f, 1e-3f, 3);
    u32x4 *w0h = dev_random_halves<u32x4>(128 * 32, -1.f, 1.f, 4), *w0l = dev_random_halves<u32x4>(128 * 32, -1e-3f, 1e-3f, 5);
    u32x4 *wfh = dev_random_halves<u32x4>((size_t)1024 * 11520, -1.f, 1.f, 6), *wfl = dev_random_halves<u32x4>((size_t)1024 * 11520, -1e-3f, 1e-3f, 7);
    float *bias, *inv, *inv_o, *part; SplitScales *sc; unsigned long long *side;
    CK(hipMalloc((void **)&bias, 1024 * 4)); CK(hipMemset(bias, 0, 1024 * 4));
    CK(hipMalloc((void **)&inv, (size_t)n * 4)); CK(hipMalloc((void **)&inv_o, (size_t)n * 4));
    CK(hipMalloc((void **)&part, (size_t)(n + 128) * 1024 * 4));
    float *fcq, *headw; unsigned *fsync; int32_t *ferr;
    CK(hipMalloc((void **)&fcq, (size_t)n * 256 * 4)); CK(hipMalloc((void **)&headw, 8 * 1024 * 4)); CK(hipMemset(headw, 0, 8 * 1024 * 4));
    const int npairs = fc_sync_pairs(n);
    CK(hipMalloc((void **)&fsync, (size_t)2 * npairs * 4)); CK(hipMemset(fsync, 0, (size_t)2 * npairs * 4));
    CK(hipMalloc((void **)&ferr, 4)); CK(hipMemset(ferr, 0, 4));
    const int heads_all = argc > 2 ? atoi(argv[2]) : 0;
    const FcFinish fin = {bias, headw, fcq, nullptr, part, fsync, fsync + npairs, ferr, heads_all};
    std::vector<float> ones(n, 1.0f / 16384.f); CK(hipMemcpy(inv, ones.data(), n * 4, hipMemcpyHostToDevice));
    SplitScales hs = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f / 16384.f, 1.f, 1.f / 16384.f, 1.f, 1.f / 16384.f};
    CK(hipMalloc((void **)&sc, sizeof hs)); CK(hipMemcpy(sc, &hs, sizeof hs, hipMemcpyHostToDevice));
    std::vector<unsigned long long> hside(2 * n);
    for (int i = 0; i < n; ++i) { hside[2 * i] = (0x0000040810204081ULL * (i % 7 + 1)) & 0xFDFBF7EFDFBFULL; hside[2 * i + 1] = (0x0000081020408102ULL << (i % 3)) & 0xFDFBF7EFDFBFULL & ~hside[2 * i]; }
    CK(hipMalloc((void **)&side, 16 * n)); CK(hipMemcpy(side, hside.data(), 16 * n, hipMemcpyHostToDevice));
    CK(prepare_net_kernels_split());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time_it = [&](auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
        return ms / iters * 1000.f;
    };
    auto block_stats = [&](int nb) {   // how the blocks of the last launch were spread over time
        std::vector<unsigned long long> bt(2 * 8192);
        CK(hipMemcpyFromSymbol(bt.data(), HIP_SYMBOL(g_sp_block_t), bt.size() * 8));
        if (nb > 8192) nb = 8192;
        unsigned long long t0 = ~0ull, t1 = 0; double dur = 0;
        for (int i = 0; i < nb; ++i) { t0 = bt[2 * i] < t0 ? bt[2 * i] : t0; t1 = bt[2 * i + 1] > t1 ? bt[2 * i + 1] : t1; dur += (double)(bt[2 * i + 1] - bt[2 * i]); }
        printf("  blocks %d: launch span %.1f us, mean block duration %.1f us, mean blocks in flight %.1f; block 100 %.1f us;", nb, (t1 - t0) / 100.0, dur / nb / 100.0, dur / (double)(t1 - t0), (bt[201] - bt[200]) / 100.0);
        std::vector<std::pair<unsigned long long, unsigned long long>> v;
        for (int i = 0; i < nb; ++i) v.push_back({bt[2 * i], bt[2 * i + 1] - bt[2 * i]});
        std::sort(v.begin(), v.end());
        printf(" by start order:");
        for (int q = 0; q < 8; ++q) { double s2 = 0; int c = 0; for (int i = q * nb / 8; i < (q + 1) * nb / 8; ++i) { s2 += v[i].second; ++c; } printf(" %.0f", s2 / c / 100.0); }
        printf(" us; by blockIdx mod 8:");
        for (int x = 0; x < 8; ++x) { double s2 = 0; int c = 0; for (int i = x; i < nb; i += 8) { s2 += bt[2 * i + 1] - bt[2 * i]; ++c; } printf(" %.0f", s2 / c / 100.0); }
        printf(" us; blocks in flight by eighth of the span:");
        for (int q = 0; q < 8; ++q) {
            const double a0 = t0 + (t1 - t0) * (q / 8.0), a1 = t0 + (t1 - t0) * ((q + 1) / 8.0); double busy = 0;
            for (int i = 0; i < nb; ++i) { double lo = fmax(a0, (double)bt[2 * i]), hi = fmin(a1, (double)bt[2 * i + 1]); if (hi > lo) busy += hi - lo; }
            printf(" %.0f", busy / (a1 - a0));
        }
        printf("\n");
    };
    {   // all three convolutions: one board per block, 16-cell tiles, two blocks per CU
        CK(hipMemset(act3, 0, (size_t)n * 90 * 512));
        float t = time_it([&] {
            conv_board_split_kernel<4, false><<<n, 256, conv_board_lds_bytes(), 0>>>(
                side, w0h, w0l, bias, w2h, w2l, bias, w2h, w2l, bias, sc, nullptr, nullptr, act3, inv_o, nullptr, n, nullptr);
        });
        // a checksum of the first boards' outputs, so that two builds of the kernel can be compared
        const int nc = n < 64 ? n : 64;
        std::vector<float> got((size_t)nc * 90 * 128);
        float *f32; CK(hipMalloc((void **)&f32, got.size() * 4));
        launch_split_to_f32(0, act3, inv_o, 90, nc, f32);
        CK(hipMemcpy(got.data(), f32, got.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0.0, big = 0.0;
        for (float v : got) { sum += v; big = fmax(big, fabs((double)v)); }
        unsigned long long h[8]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp_clk), sizeof h));
        block_stats(n);
        printf("convolutions (2->128, 7x8->8x9, 8x9->9x10), one board per block: %8.1f us  %7.1f TFLOP/s   block: 2->128 %llu  main1 %llu  epilogue1 %llu  main2 %llu  epilogue2 %llu  total %llu cycles  clock %.0f MHz;  output sum %.9g max %.6g\n",
               t, (double)n * 2 * 128 * (72 + 90) * 2048 / (t * 1e-6) / 1e12, h[1] - h[0], h[2] - h[1], h[4] - h[2], h[7] - h[4], h[3] - h[7], h[3] - h[0],
               (double)(h[3] - h[0]) / (double)(h[6] - h[5]) * 100.0, sum, big);
    }
    auto duo_run = [&](auto TGc) {       // two boards per block, pixels-only images: must reproduce the checksum above
        constexpr int TG = decltype(TGc)::value;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_duo_split_kernel<TG, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_duo_lds_bytes()));
        CK(hipMemset(act3, 0, (size_t)n * 90 * 512));
        float t = time_it([&] {
            conv_duo_split_kernel<TG, false><<<(n + 1) / 2, 256, conv_duo_lds_bytes(), 0>>>(
                side, w0h, w0l, bias, w2h, w2l, bias, w2h, w2l, bias, sc, nullptr, nullptr, act3, inv_o, nullptr, n, nullptr);
        });
        const int nc = n < 64 ? n : 64;
        std::vector<float> got((size_t)nc * 90 * 128);
        float *f32; CK(hipMalloc((void **)&f32, got.size() * 4));
        launch_split_to_f32(0, act3, inv_o, 90, nc, f32);
        CK(hipMemcpy(got.data(), f32, got.size() * 4, hipMemcpyDeviceToHost));
        double sum = 0.0, big = 0.0;
        for (float v : got) { sum += v; big = fmax(big, fabs((double)v)); }
        unsigned long long h[8]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp_clk), sizeof h));
        block_stats((n + 1) / 2);
        printf("convolutions, two boards per block (tile groups of %d):           %8.1f us  %7.1f TFLOP/s   block: 2->128 %llu  main1 %llu  epilogue1 %llu  main2 %llu  epilogue2 %llu  total %llu cycles  clock %.0f MHz;  output sum %.9g max %.6g\n",
               TG, t, (double)n * 2 * 128 * (72 + 90) * 2048 / (t * 1e-6) / 1e12, h[1] - h[0], h[2] - h[1], h[4] - h[2], h[7] - h[4], h[3] - h[7], h[3] - h[0],
               (double)(h[3] - h[0]) / (double)(h[6] - h[5]) * 100.0, sum, big);
    };
    duo_run(std::integral_constant<int, 3>{});
    duo_run(std::integral_constant<int, 4>{});
    duo_run(std::integral_constant<int, 6>{});
    u32x4 *a3 = dev_random_halves<u32x4>((size_t)n * 90 * 256, 0.f, 1.f, 8);
    auto fc_run = [&](auto BRc, auto ABLc, int rows) {
        constexpr int BR = decltype(BRc)::value, ABL = decltype(ABLc)::value;
        float t = time_it([&] {
            int groups = cdiv_sp(cdiv_sp(rows, 128), 32);
            fc_split_kernel<BR, ABL><<<256 * groups, 256, 0, 0>>>(a3, inv, wfh, wfl, sc, fin, nullptr, rows, n, 0, nullptr);
        });
        unsigned long long h[8]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp_clk), sizeof h));
        printf("linear 11520->1024, %4d rows, ring %d, ablate %2d: %8.1f us  %7.1f TFLOP/s  block %llu cycles, clock %.0f MHz\n", rows, BR, ABL, t,
               (double)rows * 2 * 11520 * 1024 / (t * 1e-6) / 1e12, h[3] - h[0], (double)(h[3] - h[0]) / (double)(h[6] - h[5]) * 100.0);
    };
    using I0 = std::integral_constant<int, 0>;
    {   // the two convolution forms as a self-play chain runs them: each launch between two launches of the linear layer
        // (whose 47 MB of weights pass through the L2s in between), every launch timed by its own pair of events
        std::vector<hipEvent_t> ev(2 * iters + 2);
        for (auto &e : ev) CK(hipEventCreate(&e));
        const int groups = cdiv_sp(cdiv_sp(n, 128), 32);
        for (int rep = 0; rep < 2; ++rep)
            for (int form = 0; form < 2; ++form) {
                for (int i = 0; i < iters + 1; ++i) {
                    fc_split_kernel<8, 0><<<256 * groups, 256, 0, 0>>>(a3, inv, wfh, wfl, sc, fin, nullptr, n, n, 0, nullptr);
                    CK(hipEventRecord(ev[2 * i], 0));
                    if (form == 0)
                        conv_board_split_kernel<4, false><<<n, 256, conv_board_lds_bytes(), 0>>>(
                            side, w0h, w0l, bias, w2h, w2l, bias, w2h, w2l, bias, sc, nullptr, nullptr, act3, inv_o, nullptr, n, nullptr);
                    else
                        conv_duo_split_kernel<4, false><<<(n + 1) / 2, 256, conv_duo_lds_bytes(), 0>>>(
                            side, w0h, w0l, bias, w2h, w2l, bias, w2h, w2l, bias, sc, nullptr, nullptr, act3, inv_o, nullptr, n, nullptr);
                    CK(hipEventRecord(ev[2 * i + 1], 0));
                }
                CK(hipDeviceSynchronize());
                double tot = 0, lo = 1e9, hi = 0;
                for (int i = 1; i < iters + 1; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; lo = fmin(lo, ms); hi = fmax(hi, ms); }
                printf("between linear layers, %s: mean %.1f us, fastest %.1f, slowest %.1f\n", form ? "two boards per block" : "one board per block ", tot / iters * 1000, lo * 1000, hi * 1000);
            }
    }
    fc_run(std::integral_constant<int, 8>{}, I0{}, n);
    fc_run(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{}, n);
    fc_run(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{}, n);
    fc_run(std::integral_constant<int, 8>{}, std::integral_constant<int, 5>{}, n);
    fc_run(std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{}, n);
    fc_run(std::integral_constant<int, 8>{}, std::integral_constant<int, 15>{}, n);
    {   // the linear layer's main loop on 16x16x32 tiles (tools/fc16_experiment.hip): timing only
        float *sink; CK(hipMalloc((void **)&sink, (size_t)1024 * 256 * 4));
        auto fc16_run = [&](auto BRc, auto ABLc) {
            constexpr int BR = decltype(BRc)::value, ABL = decltype(ABLc)::value;
            float t = time_it([&] {
                int groups = cdiv_sp(cdiv_sp(n, 128), 32);
                fc16_proto_kernel<BR, ABL><<<256 * groups, 256, 0, 0>>>(a3, wfh, wfl, sink, n, n);
            });
            unsigned long long h[8]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp_clk), sizeof h));
            printf("linear 11520->1024 main loop on 16x16x32, %4d rows, ring %d, ablate %2d: %8.1f us  %7.1f TFLOP/s  block %llu cycles, clock %.0f MHz\n", n, BR, ABL, t,
                   (double)n * 2 * 11520 * 1024 / (t * 1e-6) / 1e12, h[3] - h[0], (double)(h[3] - h[0]) / (double)(h[6] - h[5]) * 100.0);
        };
        auto fc16_product = [&](int rows) {          // the product's linear layer (fc16_kernel), epilogue and hand-off included
            float t = time_it([&] {
                int groups = cdiv_sp(cdiv_sp(rows, 128), 32);
                fc16_kernel<FC16_BR><<<256 * groups, 256, 0, 0>>>(a3, inv, wfh, wfl, sc, fin, nullptr, rows, n, 0, nullptr);
            });
            unsigned long long h[8]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp_clk), sizeof h));
            printf("linear 11520->1024 on 16x16x32 (fc16_kernel), %4d rows: %8.1f us  %7.1f TFLOP/s  block %llu cycles, clock %.0f MHz\n", rows, t,
                   (double)rows * 2 * 11520 * 1024 / (t * 1e-6) / 1e12, h[3] - h[0], (double)(h[3] - h[0]) / (double)(h[6] - h[5]) * 100.0);
        };
        fc16_product(n);
        fc16_product(n / 2);
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        fc16_run(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{});
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{});
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 8>{});
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 16>{});
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 31>{});
        fc_run(std::integral_constant<int, 6>{}, I0{}, n);
        fc16_run(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
    }
    fc_run(std::integral_constant<int, 6>{}, I0{}, n / 2);
    fc_run(std::integral_constant<int, 6>{}, I0{}, n);
    { int herr = 0; CK(hipMemcpy(&herr, ferr, 4, hipMemcpyDeviceToHost)); printf("linear layer: %d waits between partner blocks ran out\n", herr); }
    return 0;
}

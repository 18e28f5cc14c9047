// Timing experiments on the f16 conv kernel (not part of the product): ablation variants
// of conv_mid_f16_kernel on synthetic data, timed with HIP events.
#define NNB_CONV_BENCH_CLOCKS 1
#include "../alphazero_amd/csrc/net_kernels_f16.hip"
#include <cstdio>
#include <vector>
#include <cstring>
using namespace nnb;
static double g_last_mhz = 0, g_last_cycles = 0;

template <int IH, int IW, int NB, int WPS, int BRING, int ARING, int ABL, int NW = 1>
float run(const _Float16 *in, const _Float16 *wp, const float *bias, _Float16 *out, int n, int iters) {
    constexpr size_t lds = conv_lds_bytes(IH, IW, NB);
    auto k = conv_mid_f16_kernel<IH, IW, NB, WPS, BRING, ARING, false, ABL, NW>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<(n + NB - 1) / NB, 256, lds, 0>>>(in, wp, bias, out, nullptr, n, nullptr, nullptr, nullptr);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) k<<<(n + NB - 1) / NB, 256, lds, 0>>>(in, wp, bias, out, nullptr, n, nullptr, nullptr, nullptr);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    unsigned long long h[2]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_clk), 16);
    g_last_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0; g_last_cycles = (double)h[0];
    return ms / iters * 1000.f;
}

template <int IH, int IW, int NB, int MH, int BRING, int ARING>
float run_rows(const _Float16 *in, const _Float16 *wp, const float *bias, _Float16 *out, int n, int iters) {
    constexpr size_t lds = conv_rows_lds_bytes(IH, IW, NB);
    auto k = conv_rows_f16_kernel<IH, IW, NB, MH, BRING, ARING, false>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<(n + NB - 1) / NB, 256 * MH, lds, 0>>>(in, wp, bias, out, nullptr, n, nullptr, nullptr, nullptr);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) k<<<(n + NB - 1) / NB, 256 * MH, lds, 0>>>(in, wp, bias, out, nullptr, n, nullptr, nullptr, nullptr);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / iters * 1000.f;
}

static const unsigned long long *g_side = nullptr;
static const float *g_w0 = nullptr, *g_b0 = nullptr;

template <int NB, int MH, int BRING, int ARING>
float run_rows_fused(const _Float16 *, const _Float16 *wp, const float *bias, _Float16 *out, int n, int iters) {
    constexpr size_t lds = conv_rows_lds_bytes(7, 8, NB);
    auto k = conv_rows_f16_kernel<7, 8, NB, MH, BRING, ARING, true>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<(n + NB - 1) / NB, 256 * MH, lds, 0>>>(nullptr, wp, bias, out, nullptr, n, g_side, g_w0, g_b0);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) k<<<(n + NB - 1) / NB, 256 * MH, lds, 0>>>(nullptr, wp, bias, out, nullptr, n, g_side, g_w0, g_b0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / iters * 1000.f;
}

float run_mid_fused(const _Float16 *, const _Float16 *wp, const float *bias, _Float16 *out, int n, int iters) {
    constexpr size_t lds = conv_lds_bytes(7, 8, 4);
    auto k = conv_mid_f16_kernel<7, 8, 4, 2, 4, 4, true>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<(n + 3) / 4, 256, lds, 0>>>(nullptr, wp, bias, out, nullptr, n, g_side, g_w0, g_b0);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) k<<<(n + 3) / 4, 256, lds, 0>>>(nullptr, wp, bias, out, nullptr, n, g_side, g_w0, g_b0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / iters * 1000.f;
}

template <int ABL, int BR = 8, int NSET = 2>
float run_fc(const _Float16 *a, const _Float16 *wp, float *part, int n, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fc_f16_kernel<ABL, BR, NSET><<<256, 256, 0, 0>>>(a, wp, part, nullptr, n, n);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) fc_f16_kernel<ABL, BR, NSET><<<256, 256, 0, 0>>>(a, wp, part, nullptr, n, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_clk), 16);
    g_last_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0; g_last_cycles = (double)h[0];
    return ms / iters * 1000.f;
}

int main(int argc, char **argv) {
    const int n = 4096;
    _Float16 *in, *wp, *out; float *bias;
    hipMalloc(&in, (size_t)n * 72 * 128 * 2); hipMalloc(&out, (size_t)n * 90 * 128 * 2);
    hipMalloc(&wp, 128 * 128 * 16 * 2); hipMalloc(&bias, 512);
    std::vector<_Float16> h((size_t)n * 72 * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(wp, h.data(), 128 * 128 * 16 * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 512);
    const double f2 = 2.0 * 128 * 72 * 2048 * n, f4 = 2.0 * 128 * 90 * 2048 * n;
    {   // random legal-looking side boards and 2->128 weights for the fused-input variants
        std::vector<unsigned long long> sb(2 * (size_t)n);
        unsigned long long x = 88172645463325252ULL;
        for (size_t i = 0; i < sb.size(); i += 2) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            unsigned long long a = x & 0xFDFBF7EFDFBFULL; x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            sb[i] = a & x; sb[i + 1] = (x >> 3) & 0xFDFBF7EFDFBFULL & ~sb[i];
        }
        unsigned long long *dsb; float *w0, *b0;
        hipMalloc(&dsb, sb.size() * 8); hipMemcpy(dsb, sb.data(), sb.size() * 8, hipMemcpyHostToDevice);
        std::vector<float> hw(128 * 32 + 128);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u >> 20) & 255) / 512.f - 0.2f;
        hipMalloc(&w0, hw.size() * 4); hipMemcpy(w0, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        g_side = dsb; g_w0 = w0; g_b0 = w0 + 128 * 32;
    }
    struct Var { const char *name; float (*fn)(const _Float16 *, const _Float16 *, const float *, _Float16 *, int, int); double flop; float best; double mhz, cyc; };
#define C4(NB, WPS, BR, AR, ABL, NW) {"conv4 NB=" #NB " wps=" #WPS " bring=" #BR " aring=" #AR " abl=" #ABL " NW=" #NW, run<8, 9, NB, WPS, BR, AR, ABL, NW>, f4, 1e9f}
#define C2(NB, WPS, BR, AR, ABL, NW) {"conv2 NB=" #NB " wps=" #WPS " bring=" #BR " aring=" #AR " abl=" #ABL " NW=" #NW, run<7, 8, NB, WPS, BR, AR, ABL, NW>, f2, 1e9f}
    std::vector<Var> vars = {
        C4(2, 2, 4, 12, 0, 1), C2(4, 2, 4, 4, 0, 1),
#define Q4(NB, MH, BR, AR) {"conv4 rows NB=" #NB " MH=" #MH " bring=" #BR " aring=" #AR, run_rows<8, 9, NB, MH, BR, AR>, f4, 1e9f}
#define Q2(NB, MH, BR, AR) {"conv2 rows NB=" #NB " MH=" #MH " bring=" #BR " aring=" #AR, run_rows<7, 8, NB, MH, BR, AR>, f2, 1e9f}
        Q4(3, 1, 4, 8), Q4(3, 1, 4, 4), Q4(3, 1, 2, 8), Q4(3, 1, 8, 8), Q4(6, 2, 4, 8),
        Q2(7, 2, 4, 8), Q2(3, 1, 4, 8),
        {"conv2 fused cell tiles NB=4 (product)", run_mid_fused, f2, 1e9f},
        {"conv2 fused rows NB=7 MH=2 aring 8", run_rows_fused<7, 2, 4, 8>, f2, 1e9f},
        {"conv2 fused rows NB=7 MH=2 aring 4", run_rows_fused<7, 2, 4, 4>, f2, 1e9f},
        {"conv2 fused rows NB=3 MH=1 aring 8", run_rows_fused<3, 1, 4, 8>, f2, 1e9f},
    };
    {   // the row-tile kernels must reproduce the cell-tile kernels bit for bit
        auto check = [&](const char *name, auto ref_fn, auto new_fn, size_t halves, int nn) {
            std::vector<_Float16> a(halves), b(halves);
            hipMemset(out, 0xFF, halves * 2); ref_fn(in, wp, bias, out, nn, 1); hipMemcpy(a.data(), out, halves * 2, hipMemcpyDeviceToHost);
            hipMemset(out, 0xFF, halves * 2); new_fn(in, wp, bias, out, nn, 1); hipMemcpy(b.data(), out, halves * 2, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < halves; ++i) if (memcmp(&a[i], &b[i], 2)) { if (!bad) first = i; ++bad; }
            printf("check %-28s n=%d: %zu of %zu halves differ (first at %zu)\n", name, nn, bad, halves, first);
        };
        for (int nn : {4096, 4093, 5}) {
            check("conv4 rows NB=3", run<8, 9, 2, 2, 4, 12, 0, 1>, run_rows<8, 9, 3, 1, 4, 8>, (size_t)nn * 90 * 128, nn);
            check("conv4 rows NB=6 MH=2", run<8, 9, 2, 2, 4, 12, 0, 1>, run_rows<8, 9, 6, 2, 4, 8>, (size_t)nn * 90 * 128, nn);
            check("conv2 rows NB=7 MH=2", run<7, 8, 4, 2, 4, 4, 0, 1>, run_rows<7, 8, 7, 2, 4, 8>, (size_t)nn * 72 * 128, nn);
            check("conv2 rows NB=3", run<7, 8, 4, 2, 4, 4, 0, 1>, run_rows<7, 8, 3, 1, 4, 8>, (size_t)nn * 72 * 128, nn);
            check("conv2 fused rows NB=7 MH=2", run_mid_fused, run_rows_fused<7, 2, 4, 8>, (size_t)nn * 72 * 128, nn);
            check("conv2 fused rows NB=3", run_mid_fused, run_rows_fused<3, 1, 4, 8>, (size_t)nn * 72 * 128, nn);
        }
    }
    for (int rep = 0; rep < 5; ++rep)
        for (auto &v : vars) { float us = v.fn(in, wp, bias, out, n, 10); if (us < v.best) { v.best = us; v.mhz = g_last_mhz; v.cyc = g_last_cycles; } }
    for (auto &v : vars) printf("%-48s %8.1f us %7.1f TF  block 700: %.0f cycles at %.0f MHz\n", v.name, v.best, v.flop / v.best * 1e-6, v.cyc, v.mhz);
    {
        _Float16 *fa, *fw; float *part;
        hipMalloc(&fa, (size_t)n * 11520 * 2); hipMalloc(&fw, (size_t)1024 * 11520 * 2); hipMalloc(&part, (size_t)2 * n * 1024 * 4);
        std::vector<_Float16> hh((size_t)n * 11520);
        for (size_t i = 0; i < hh.size(); ++i) hh[i] = (_Float16)((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
        hipMemcpy(fa, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(fw, hh.data(), (size_t)1024 * 11520 * 2, hipMemcpyHostToDevice);
        const double ff = 2.0 * 11520 * 1024 * n;
#define RF(ABL) { float us = run_fc<ABL>(fa, fw, part, n, 20); printf("fc    ablate=%2d %8.1f us %7.1f TF  main loop of block 100: %.0f cycles at %.0f MHz\n", ABL, us, ff / us * 1e-6, g_last_cycles, g_last_mhz); }
        RF(0) RF(0)
#define RG(BR, NSET) { float us = run_fc<0, BR, NSET>(fa, fw, part, n, 20); printf("fc    BR=%2d NSET=%d %8.1f us %7.1f TF  main loop of block 100: %.0f cycles at %.0f MHz\n", BR, NSET, us, ff / us * 1e-6, g_last_cycles, g_last_mhz); }
        RF(0)
    }
    return 0;
}

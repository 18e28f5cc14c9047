// Timing experiments on the f16 conv kernel (not part of the product): ablation variants
// of conv_mid_f16_kernel on synthetic data, timed with HIP events.
#include "../alphazero_amd/csrc/net_kernels_f16.hip"
#include <cstdio>
#include <vector>
using namespace nnb;

template <int IH, int IW, int NB, int WPS, int BRING, int ARING, int ABL>
float run(const _Float16 *in, const _Float16 *wp, const float *bias, _Float16 *out, int n, int iters) {
    constexpr size_t lds = conv_lds_bytes(IH, IW, NB);
    auto k = conv_mid_f16_kernel<IH, IW, NB, WPS, BRING, ARING, false, ABL>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<(n + NB - 1) / NB, 256, lds, 0>>>(in, wp, bias, out, nullptr, n, nullptr, nullptr, nullptr);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) k<<<(n + NB - 1) / NB, 256, lds, 0>>>(in, wp, bias, out, nullptr, n, nullptr, nullptr, nullptr);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / iters * 1000.f;
}

template <int ABL>
float run_fc(const _Float16 *a, const _Float16 *wp, float *part, int n, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fc_f16_kernel<ABL><<<256, 256, 0, 0>>>(a, wp, part, nullptr, n, n);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) fc_f16_kernel<ABL><<<256, 256, 0, 0>>>(a, wp, part, nullptr, n, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
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
#define R2(ABL) { float us = run<7, 8, 4, 2, 4, 8, ABL>(in, wp, bias, out, n, 20); printf("conv2 ablate=%2d %8.1f us %7.1f TF\n", ABL, us, f2 / us * 1e-6); }
#define R4(ABL) { float us = run<8, 9, 2, 2, 4, 8, ABL>(in, wp, bias, out, n, 20); printf("conv4 ablate=%2d %8.1f us %7.1f TF\n", ABL, us, f4 / us * 1e-6); }
    R2(0) R2(15)
    R4(0) R4(15)
#define V4(NB, WPS, BR, AR) { float us = run<8, 9, NB, WPS, BR, AR, 0>(in, wp, bias, out, n, 20); printf("conv4 NB=%d wps=%d bring=%d aring=%d %8.1f us %7.1f TF\n", NB, WPS, BR, AR, us, f4 / us * 1e-6); }
    V4(2, 3, 4, 6) V4(2, 3, 2, 6) V4(2, 3, 2, 4) V4(2, 2, 4, 12) V4(2, 2, 8, 8) V4(4, 1, 8, 16) V4(4, 1, 4, 8) V4(3, 2, 4, 8)
#define V2(NB, WPS, BR, AR) { float us = run<7, 8, NB, WPS, BR, AR, 0>(in, wp, bias, out, n, 20); printf("conv2 NB=%d wps=%d bring=%d aring=%d %8.1f us %7.1f TF\n", NB, WPS, BR, AR, us, f2 / us * 1e-6); }
    V2(4, 2, 2, 8) V2(4, 2, 4, 4) V2(4, 2, 4, 12) V2(8, 1, 8, 16) V2(8, 1, 4, 8)
    {
        _Float16 *fa, *fw; float *part;
        hipMalloc(&fa, (size_t)n * 11520 * 2); hipMalloc(&fw, (size_t)1024 * 11520 * 2); hipMalloc(&part, (size_t)2 * n * 1024 * 4);
        std::vector<_Float16> hh((size_t)n * 11520);
        for (size_t i = 0; i < hh.size(); ++i) hh[i] = (_Float16)((float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f);
        hipMemcpy(fa, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(fw, hh.data(), (size_t)1024 * 11520 * 2, hipMemcpyHostToDevice);
        const double ff = 2.0 * 11520 * 1024 * n;
#define RF(ABL) { float us = run_fc<ABL>(fa, fw, part, n, 20); printf("fc    ablate=%2d %8.1f us %7.1f TF\n", ABL, us, ff / us * 1e-6); }
        RF(0) RF(1) RF(2) RF(4) RF(5) RF(7)
    }
    return 0;
}

// Practical MFMA ceiling of the box: every SIMD issues v_mfma_f32_32x32x16_f16 back to back from
// registers (no memory traffic).  Prints achieved TFLOP/s and the shader clock under that load
// (s_memtime ticks per wall-clock tick).   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void spin(int iters, float *out, unsigned long long *clk) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    float16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// 32x32x16 again, but every MFMA takes another pair of operands filled with pseudo-random halves in [-1, 1): what the
// power management allows when the multipliers see changing data, as they do in a real kernel
__global__ __launch_bounds__(256) void spin_random(int iters, float *out, unsigned long long *clk) {
    half8 a[4], b[4];
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; a[j][i] = (_Float16)((float)(int)(s >> 8) / 8388608.0f - 1.0f);
            s = s * 1664525u + 1013904223u; b[j][i] = (_Float16)((float)(int)(s >> 8) / 8388608.0f - 1.0f);
        }
    float16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a[1]), "v"(b[1]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a[2]), "v"(b[2]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a[3]), "v"(b[3]));
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// Does it matter for the power how the operands change from one instruction to the next?  ORDER 0: both operands change
// every time (a0 b0, a1 b1, a2 b2, a3 b3).  ORDER 1: the 4 x 4 products in row order (a0 b0, a0 b1, a0 b2, a0 b3, a1 b0, ...:
// B changes every time, A every fourth).  ORDER 2: the same products in snake order (a0 b0, a0 b1, a0 b2, a0 b3, a1 b3,
// a1 b2, ...: exactly one operand changes per step).  16 accumulators, pseudo-random operands.
template <int ORDER>
__global__ __launch_bounds__(256) void spin_order(int iters, float *out, unsigned long long *clk) {
    half8 a[4], b[4];
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; a[j][i] = (_Float16)((float)(int)(s >> 8) / 8388608.0f - 1.0f);
            s = s * 1664525u + 1013904223u; b[j][i] = (_Float16)((float)(int)(s >> 8) / 8388608.0f - 1.0f);
        }
    float4v c[16];
    for (int j = 0; j < 16; ++j) c[j] = float4v{0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = ORDER == 0 ? k & 3 : k >> 2;
            const int j = ORDER == 0 ? k & 3 : ORDER == 1 ? k & 3 : ((k >> 2) & 1 ? 3 - (k & 3) : k & 3);
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a[i]), "v"(b[j]));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float r = 0;
    for (int j = 0; j < 16; ++j) r += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// same FLOPs per iteration with the 16x16x32 instruction (8 accumulators of 4 registers; inline assembly, because the
// compiler shuttles the builtin's accumulators through accumulation registers inside the loop)
__global__ __launch_bounds__(256) void spin16(int iters, float *out, unsigned long long *clk) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c5) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c6) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c7) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c0[j] + c1[j] + c2[j] + c3[j] + c4[j] + c5[j] + c6[j] + c7[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main(int argc, char **argv) {
    // 2 blocks of 4 waves per CU = 2 waves per SIMD; iterations per launch from the command line (1 000 000 = 65 ms at
    // 2 GHz: long enough for the power management to settle)
    const int blocks = 256 * 2, iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        spin<<<blocks, 256>>>(iters, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 4 * 32768.0;
        printf("rep %d: %.3f ms  %.1f TFLOP/s  shader ticks %llu  wall ticks (100 MHz) %llu -> %.0f MHz\n", rep, ms,
               flop / (ms * 1e-3) / 1e12, h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
    }
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        spin16<<<blocks, 256>>>(iters, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 8 * 16384.0;
        printf("16x16x32 rep %d: %.3f ms  %.1f TFLOP/s  shader ticks %llu  wall ticks %llu -> %.0f MHz\n", rep, ms,
               flop / (ms * 1e-3) / 1e12, h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
    }
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        spin_random<<<blocks, 256>>>(iters, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 4 * 32768.0;
        printf("32x32x16, changing pseudo-random operands, rep %d: %.3f ms  %.1f TFLOP/s -> %.0f MHz\n", rep, ms, flop / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 100.0);
    }
    auto order_run = [&](auto kernel, const char *what) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            kernel<<<blocks, 256>>>(iters / 2, out, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double flop = (double)blocks * 4 * (iters / 2) * 16 * 16384.0;
            printf("16x16x32, pseudo-random operands, %s, rep %d: %.3f ms  %.1f TFLOP/s\n", what, rep, ms, flop / (ms * 1e-3) / 1e12);
        }
    };
    order_run(spin_order<0>, "both operands change every instruction");
    order_run(spin_order<1>, "4 x 4 products in row order");
    order_run(spin_order<2>, "4 x 4 products in snake order");
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        spin<<<blocks, 256>>>(iters, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 4 * 32768.0;
        printf("32x32x16 again rep %d: %.3f ms  %.1f TFLOP/s -> %.0f MHz\n", rep, ms, flop / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 100.0);
    }
    return 0;
}

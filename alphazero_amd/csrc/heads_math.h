// The two heads of the policy-value network (network.py:40-41,48-49) and the legal-move
// mask / renormalisation of agent_code.py:199-208, as device functions shared by the
// stand-alone heads kernels and by the tree step, which finishes the forward pass of its own
// pending leaf itself.  Every function is a fixed sequence of IEEE operations (explicit fmaf,
// no contractible expressions), so all callers agree bit for bit whatever floating-point
// contraction setting their translation unit is compiled with.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnb {

struct HeadsDev {
    const float *x0;        // [rows][1024] linear-layer output, or the first split-K partial sum
    const float *x1;        // second partial sum, or null when x0 is already bias + ReLU
    const float *fc_bias;   // [1024] added before the ReLU when x1 != null
    const float *hw;        // [8][1024] policy rows 0..6, value row 7
    const float *hb;        // [8]
    int32_t x1_row_offset;  // rows between x0 and x1 are laid out as x1 = x0 + x1_row_offset * 1024
};

// Lane `lane` of a wave takes elements 4*(lane + 64*i) .. +3, i = 0..3, of row `row`.
__device__ __forceinline__ void heads_load_row(const HeadsDev &h, size_t row, int lane, float4 (&xv)[4]) {
    const float4 *p0 = reinterpret_cast<const float4 *>(h.x0 + row * 1024);
    if (h.x1) {
        const float4 *p1 = reinterpret_cast<const float4 *>(h.x1 + row * 1024);
        const float4 *fb = reinterpret_cast<const float4 *>(h.fc_bias);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 u = p0[lane + 64 * i], v = p1[lane + 64 * i], b = fb[lane + 64 * i];
            xv[i].x = fmaxf((u.x + v.x) + b.x, 0.f);
            xv[i].y = fmaxf((u.y + v.y) + b.y, 0.f);
            xv[i].z = fmaxf((u.z + v.z) + b.z, 0.f);
            xv[i].w = fmaxf((u.w + v.w) + b.w, 0.f);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = p0[lane + 64 * i];
    }
}

// Dot product of the row with one head row (1024 floats at `w`, global or LDS); the full sum
// ends up in every lane.
__device__ __forceinline__ float heads_dot(const float4 (&xv)[4], const float4 *w, int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 wv = w[lane + 64 * i];
        s = __builtin_fmaf(xv[i].x, wv.x, s);
        s = __builtin_fmaf(xv[i].y, wv.y, s);
        s = __builtin_fmaf(xv[i].z, wv.z, s);
        s = __builtin_fmaf(xv[i].w, wv.w, s);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    return s;
}

// log-softmax of the 7 policy logits (network.py:48).
__device__ __forceinline__ void heads_log_softmax(const float (&z)[8], float (&lp)[7]) {
    float mx = z[0];
#pragma unroll
    for (int o = 1; o < 7; ++o) mx = fmaxf(mx, z[o]);
    float sum = 0.f;
#pragma unroll
    for (int o = 0; o < 7; ++o) sum += expf(z[o] - mx);
    const float lse = mx + logf(sum);
#pragma unroll
    for (int o = 0; o < 7; ++o) lp[o] = z[o] - lse;
}

// agent_code.py:199-208: exp in float32, mask + renormalise in float64, uniform over the legal
// columns when everything legal underflowed.
__device__ __forceinline__ void heads_priors(const float (&lp)[7], unsigned legal, float (&prior)[7]) {
    double f[7], tot = 0.0;
#pragma unroll
    for (int o = 0; o < 7; ++o) {
        f[o] = ((legal >> o) & 1) ? (double)expf(lp[o]) : 0.0;
        tot += f[o];
    }
    const int nl = __popc(legal & 0x7F);
#pragma unroll
    for (int o = 0; o < 7; ++o) {
        const double pr = tot > 0.0 ? f[o] / tot : (((legal >> o) & 1) ? 1.0 / nl : 0.0);
        prior[o] = (float)pr;
    }
}

}  // namespace nnb

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
    const float *q;         // non-null: [rows][8 heads][32] per-column-tile partial dot products of the heads with
                            // ReLU(linear layer) (written by the split linear layer's last step); x0, x1, hw unused
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

// Heads from the per-column-tile partial sums q[head][tile] (tile = 32 consecutive columns of the linear layer's
// output).  A head's 32 partial sums are added as a balanced tree over the tile index: first tile ^ 16, then ^ 8,
// ^ 4, ^ 2, ^ 1 (the same shape as the sum over a tile's 32 columns that produced each q), then the head's bias.
// Every lane returns all 8 sums.
__device__ __forceinline__ void heads_z_from_q(const float *qrow, const float *hb, int lane, float (&z)[8]) {
    float4 v = reinterpret_cast<const float4 *>(qrow)[lane];       // head lane >> 3, tiles 4 * (lane & 7) .. + 3
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {                          // tile ^ 16, ^ 8, ^ 4  <->  lane ^ 4, ^ 2, ^ 1
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off);
        v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    const float s = (v.x + v.z) + (v.y + v.w);                        // tile ^ 2, then tile ^ 1
#pragma unroll
    for (int o = 0; o < 8; ++o) z[o] = __shfl(s, 8 * o) + hb[o];
}

// The value head alone (row 7 of q), same order of additions.
__device__ __forceinline__ float heads_value_from_q(const float *qrow, const float *hb, int lane) {
    float v = qrow[7 * 32 + (lane & 31)];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v + hb[7];
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

// Policy-value network kernels for gfx950: Connect4Net.forward
// (nnbattle/agents/alphazero/network.py:43-50) over a compacted batch of leaves.
//
// Activations are channels-last float32 ([board][row][col][128]).  The two 128->128
// 4x4/pad-2 convolutions and the 11520->1024 linear layer run on the f32-input matrix
// cores (v_mfma_f32_32x32x2_f32, exact f32 FMA chains), the input tile staged once
// in LDS and the weights streamed straight from L2 into registers in a layout
// repacked at load time.  The batch size lives in device memory (`count`), so the
// host never waits for the tree kernel before launching the forward pass.
#include <hip/hip_runtime.h>
#include <math.h>
#include "c4_bits.h"
#include "engine_layout.h"
#include "kernels.h"
#include "heads_math.h"

namespace nnb {

using c4::u64;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int batch_count(const int32_t *count_ptr, int n_fixed) {
    return count_ptr ? *count_ptr : n_fixed;
}

// ---------------------------------------------------------------------------
// Weight repacking (once per nnb_load_weights).
// ---------------------------------------------------------------------------
// Conv2d weight [n=128][c=128][kh=4][kw=4] -> [tap=kh*4+kw][c/8][(c%8)/4][n][c%4]:
// for one (tap, 8-channel chunk) a wave reads two contiguous 512-byte runs, and lane
// (n, h) holds the four channels 8*chunk + 4*h + {0..3} it feeds to four MFMAs.
__global__ void repack_conv_kernel(const float *__restrict__ w, float *__restrict__ wp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;      // index into wp
    if (i >= 128 * 128 * 16) return;
    int j = i & 3, n = (i >> 2) & 127, h = (i >> 9) & 1, chunk = (i >> 10) & 15, tap = i >> 14;
    int c = chunk * 8 + h * 4 + j, kh = tap >> 2, kw = tap & 3;
    wp[i] = w[((n * 128 + c) * 4 + kh) * 4 + kw];
}

// Linear weight [n=1024][k_ref = c*90 + hw] -> [kk/8][(kk%8)/4][n][kk%4] with
// kk = hw*128 + c, the flatten order of channels-last activations
// (network.py:46 flattens NCHW: c*90 + h*10 + w).
__global__ void repack_fc_kernel(const float *__restrict__ w, float *__restrict__ wp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)1024 * 11520) return;
    int j = i & 3, n = (i >> 2) & 1023, h = (i >> 12) & 1;
    int chunk = (int)(i >> 13);
    int kk = chunk * 8 + h * 4 + j, hw = kk >> 7, c = kk & 127;
    wp[i] = w[(size_t)n * 11520 + c * 90 + hw];
}

// policy_head [7][1024] + value_head [1][1024] -> [8][1024]; biases -> [8].
__global__ void repack_heads_kernel(const float *__restrict__ pw, const float *__restrict__ pb,
                                    const float *__restrict__ vw, const float *__restrict__ vb,
                                    float *__restrict__ hw, float *__restrict__ hb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 7 * 1024) hw[i] = pw[i];
    else if (i < 8 * 1024) hw[i] = vw[i - 7 * 1024];
    if (i < 7) hb[i] = pb[i];
    if (i == 7) hb[7] = vb[0];
}

__global__ void copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// float planes [n,2,6,7] (entries 0/1) -> side boards.
__global__ void planes_to_bits_kernel(int n, const float *__restrict__ planes, u64 *__restrict__ side) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n) return;
    const float *p = planes + (size_t)i * 42;
    u64 bits = 0;
    for (int row = 0; row < 6; ++row)
        for (int col = 0; col < 7; ++col)
            if (p[row * 7 + col] >= 0.5f) bits |= 1ULL << (col * 7 + (5 - row));
    side[i] = bits;
}

// ---------------------------------------------------------------------------
// K3+K4a: leaf encode fused with conv 2->128 (network.py:29-30).  The input planes
// are bits, so each output is the bias plus the weights under the set bits.
// One thread per output channel, 128 threads per block, boards grid-strided.
// out: [board][7][8][128], ReLU applied.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void conv_in_kernel(const u64 *__restrict__ side,
                                                      const float *__restrict__ w,   // [128][2][4][4]
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ out,
                                                      const int32_t *__restrict__ count_ptr, int n_fixed) {
    const int count = batch_count(count_ptr, n_fixed);
    const int n = threadIdx.x;
    float wr[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) wr[k] = w[n * 32 + k];
    const float b = bias[n];
    for (int board = blockIdx.x; board < count; board += gridDim.x) {
        // uniform across the block: keep the two boards in scalar registers
        unsigned lo0 = __builtin_amdgcn_readfirstlane((unsigned)side[2 * board]);
        unsigned hi0 = __builtin_amdgcn_readfirstlane((unsigned)(side[2 * board] >> 32));
        unsigned lo1 = __builtin_amdgcn_readfirstlane((unsigned)side[2 * board + 1]);
        unsigned hi1 = __builtin_amdgcn_readfirstlane((unsigned)(side[2 * board + 1] >> 32));
        const u64 pl[2] = {((u64)hi0 << 32) | lo0, ((u64)hi1 << 32) | lo1};
        float *o = out + (size_t)board * 56 * 128 + n;
#pragma unroll
        for (int oh = 0; oh < 7; ++oh) {
#pragma unroll
            for (int ow = 0; ow < 8; ++ow) {
                float acc = b;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 4; ++kw) {
                            const int ih = oh + kh - 2, iw = ow + kw - 2;   // reference row index
                            if (ih >= 0 && ih < 6 && iw >= 0 && iw < 7) {
                                if ((pl[ci] >> (iw * 7 + (5 - ih))) & 1) acc += wr[(ci * 4 + kh) * 4 + kw];
                            }
                        }
                o[(oh * 8 + ow) * 128] = fmaxf(acc, 0.0f);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K4b: conv 128->128, 4x4, pad 2, + bias + ReLU (network.py:31-34) as an implicit
// GEMM on v_mfma_f32_32x32x2_f32.
//   in : [board][IH][IW][128]      out: [board][IH+1][IW+1][128]
// A block owns NB boards: their inputs are staged once in LDS (16-byte chunks
// XOR-swizzled by position so the 32 rows a wave reads hit different banks) plus one
// all-zero row that out-of-range taps read.  Wave w computes output channels
// 32w..32w+31 for all NB*(IH+1)*(IW+1) output positions (MT tiles of 32 rows); its
// weight fragments come straight from global memory (each wave needs different
// weights, so LDS would add nothing) one 16-byte load per lane per 8 channels.
// ---------------------------------------------------------------------------
template <int IH, int IW, int NB>
__global__ __launch_bounds__(256, 1) void conv_mid_kernel(const float *__restrict__ in,
                                                          const float *__restrict__ wp,
                                                          const float *__restrict__ bias,
                                                          float *__restrict__ out,
                                                          const int32_t *__restrict__ count_ptr, int n_fixed) {
    constexpr int OH = IH + 1, OW = IW + 1, IP = IH * IW, OP = OH * OW;
    constexpr int M = NB * OP, MT = (M + 31) / 32, ROWS = NB * IP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *lds = reinterpret_cast<float4 *>(smem_raw);          // [(ROWS+1)][32] float4

    const int count = batch_count(count_ptr, n_fixed);
    const int b0 = blockIdx.x * NB;
    if (b0 >= count) return;
    const int nb_valid = min(NB, count - b0);
    const int tid = threadIdx.x;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *src = reinterpret_cast<const float4 *>(in + (size_t)b0 * IP * 128);
    for (int e = tid; e < ROWS * 32; e += 256) {
        int p = e >> 5, q = e & 31;
        float4 v = (p < nb_valid * IP) ? src[e] : zero4;
        lds[p * 32 + (q ^ (p & 15))] = v;
    }
    if (tid < 32) lds[ROWS * 32 + tid] = zero4;
    __syncthreads();

    const int lane = tid & 63, w = tid >> 6, m = lane & 31, h = lane >> 5;
    int pbase[MT], ohv[MT], owv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int r = t * 32 + m;
        int j = r / OP, o = r % OP;
        ohv[t] = (r < M) ? o / OW : -100;        // -100: every tap lands out of range
        owv[t] = o % OW;
        pbase[t] = j * IP + (o / OW) * IW + (o % OW);
    }
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const float4 *wlane = reinterpret_cast<const float4 *>(wp) + h * 128 + w * 32 + m;
    float4 bnext = wlane[0];
    for (int tap = 0; tap < 16; ++tap) {
        const int dh = (tap >> 2) - 2, dw = (tap & 3) - 2;
        int rowoff[MT], sw[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            int ih = ohv[t] + dh, iw = owv[t] + dw;
            bool ok = (unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW;
            int p = ok ? pbase[t] + dh * IW + dw : ROWS;
            rowoff[t] = p * 32;
            sw[t] = p & 15;
        }
#pragma unroll 2
        for (int chunk = 0; chunk < 16; ++chunk) {
            const float4 b = bnext;
            int nxt = tap * 16 + chunk + 1;
            nxt = nxt < 256 ? nxt : 255;
            bnext = wlane[nxt * 256];
            const int q = 2 * chunk + h;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float4 a = lds[rowoff[t] + (q ^ sw[t])];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
            }
        }
    }

    const float bv = bias[w * 32 + m];
    float *obase = out + (size_t)b0 * OP * 128 + w * 32 + m;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int R = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (R < M && R < nb_valid * OP) obase[(size_t)R * 128] = fmaxf(acc[t][r] + bv, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------
// K5: linear 11520->1024 + bias + ReLU (network.py:36-39,46-47) on
// v_mfma_f32_32x32x2_f32.  Block tile 128 rows x 128 columns, 4 waves, wave w owns
// columns 32w..32w+31 of the tile for all four 32-row tiles.  A slabs of 64 k are
// double-buffered in LDS (loaded to registers before the MFMAs of the current slab,
// written after them); B fragments stream from global memory in the repacked layout.
// ---------------------------------------------------------------------------
constexpr int FC_K = 11520, FC_N = 1024, FC_BM = 128, FC_BK = 64;

__global__ __launch_bounds__(256, 1) void fc_kernel(const float *__restrict__ a,      // [rows][11520]
                                                    const float *__restrict__ wp,     // repacked
                                                    const float *__restrict__ bias,
                                                    float *__restrict__ out,          // [rows][1024]
                                                    const int32_t *__restrict__ count_ptr, int n_fixed,
                                                    int max_rows) {
    __shared__ __attribute__((aligned(16))) float4 lds[2][FC_BM * 16];
    const int count = batch_count(count_ptr, n_fixed);
    // blocks that share an XCD (same blockIdx % 8) and are launched together work on
    // the same rows with different columns, so the A slabs are re-read from that L2
    const int bid = blockIdx.x;
    const int mb = (bid & 7) + 8 * (bid >> 6), nb = (bid >> 3) & 7;
    const int row0 = mb * FC_BM;
    if (row0 >= count) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 31, h = lane >> 5;

    // staging map: thread -> (row, 16-byte column) pairs of a 128 x 64 slab
    int srow[8], scol[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int e = tid + 256 * i;
        srow[i] = e >> 4; scol[i] = e & 15;
    }
    const float4 *abase[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int r = min(row0 + srow[i], max_rows - 1);
        abase[i] = reinterpret_cast<const float4 *>(a + (size_t)r * FC_K) + scol[i];
    }
    float4 stage[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) stage[i] = abase[i][0];
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[0][srow[i] * 16 + (scol[i] ^ (srow[i] & 15))] = stage[i];
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const float4 *wlane = reinterpret_cast<const float4 *>(wp) + h * FC_N + nb * 128 + w * 32 + m;
    constexpr int NSLAB = FC_K / FC_BK;      // 180
    float4 bnext = wlane[0];
    for (int s = 0; s < NSLAB; ++s) {
        const int cur = s & 1;
        if (s + 1 < NSLAB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) stage[i] = abase[i][(s + 1) * 16];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 b = bnext;
            int nxt = s * 8 + c + 1;
            nxt = nxt < NSLAB * 8 ? nxt : NSLAB * 8 - 1;
            bnext = wlane[(size_t)nxt * 2 * FC_N];
            const int q = 2 * c + h;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = t * 32 + m;
                const float4 av = lds[cur][row * 16 + (q ^ (row & 15))];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b.w, acc[t], 0, 0, 0);
            }
        }
        if (s + 1 < NSLAB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) lds[cur ^ 1][srow[i] * 16 + (scol[i] ^ (srow[i] & 15))] = stage[i];
        }
        __syncthreads();
    }

    const int col = nb * 128 + w * 32 + m;
    const float bv = bias[col];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < count) out[(size_t)row * FC_N + col] = fmaxf(acc[t][r] + bv, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------
// K6: heads (network.py:40-41,48-49): 1024->7 log-softmax and 1024->1 tanh, plus the
// legal-move mask / renormalisation of agent_code.py:199-208.  One wave per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heads_kernel(HeadsDev hd, const uint8_t *__restrict__ legal,
                                                    float *__restrict__ logp_out, float *__restrict__ value_out,
                                                    float *__restrict__ prior_out,
                                                    const int32_t *__restrict__ count_ptr, int n_fixed) {
    const int count = batch_count(count_ptr, n_fixed);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave; row < count; row += nwaves) {
        float4 xv[4];
        heads_load_row(hd, row, lane, xv);
        float z[8];
#pragma unroll
        for (int o = 0; o < 8; ++o)
            z[o] = heads_dot(xv, reinterpret_cast<const float4 *>(hd.hw + o * 1024), lane) + hd.hb[o];
        if (lane == 0) {
            float lp[7];
            heads_log_softmax(z, lp);
            if (logp_out)
#pragma unroll
                for (int o = 0; o < 7; ++o) logp_out[(size_t)row * 7 + o] = lp[o];
            if (value_out) value_out[row] = tanhf(z[7]);
            if (prior_out && legal) {
                float pr[7];
                heads_priors(lp, legal[row], pr);
#pragma unroll
                for (int o = 0; o < 7; ++o) prior_out[(size_t)row * 7 + o] = pr[o];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Host launchers
// ---------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

HeadsDev heads_input_f32(const NetWeights &w, const NetBuffers &b) {
    return HeadsDev{nullptr, b.fc, nullptr, nullptr, w.heads_w, w.heads_b, 0};
}

constexpr int CONV2_NB = 4, CONV4_NB = 2;
constexpr size_t CONV2_LDS = (size_t)(CONV2_NB * 7 * 8 + 1) * 512;
constexpr size_t CONV4_LDS = (size_t)(CONV4_NB * 8 * 9 + 1) * 512;

hipError_t prepare_net_kernels() {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mid_kernel<7, 8, CONV2_NB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV2_LDS);
    if (err != hipSuccess) return err;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mid_kernel<8, 9, CONV4_NB>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV4_LDS);
}

void launch_repack(hipStream_t st, const NetWeights &dst, const float *conv0_w, const float *conv0_b,
                   const float *conv2_w, const float *conv2_b, const float *conv4_w, const float *conv4_b,
                   const float *fc_w, const float *fc_b, const float *policy_w, const float *policy_b,
                   const float *value_w, const float *value_b) {
    copy_kernel<<<cdiv(128 * 32, 256), 256, 0, st>>>(conv0_w, dst.conv0_w, 128 * 32);
    copy_kernel<<<1, 256, 0, st>>>(conv0_b, dst.conv0_b, 128);
    if (dst.conv2_w) repack_conv_kernel<<<cdiv(128 * 128 * 16, 256), 256, 0, st>>>(conv2_w, dst.conv2_w);
    copy_kernel<<<1, 256, 0, st>>>(conv2_b, dst.conv2_b, 128);
    if (dst.conv4_w) repack_conv_kernel<<<cdiv(128 * 128 * 16, 256), 256, 0, st>>>(conv4_w, dst.conv4_w);
    copy_kernel<<<1, 256, 0, st>>>(conv4_b, dst.conv4_b, 128);
    if (dst.fc_w) repack_fc_kernel<<<(1024 * 11520) / 256, 256, 0, st>>>(fc_w, dst.fc_w);
    copy_kernel<<<4, 256, 0, st>>>(fc_b, dst.fc_b, 1024);
    repack_heads_kernel<<<32, 256, 0, st>>>(policy_w, policy_b, value_w, value_b, dst.heads_w, dst.heads_b);
}

void launch_planes_to_bits(hipStream_t st, int n, const float *planes, u64 *side) {
    if (n <= 0) return;
    planes_to_bits_kernel<<<cdiv(2 * n, 256), 256, 0, st>>>(n, planes, side);
}

void launch_net_stage(hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
                      const uint8_t *legal, float *logp_out, float *value_out, float *prior_out,
                      const int32_t *count_ptr, int n_max) {
    if (n_max <= 0) return;
    switch (stage) {
    case NET_CONV0:
        conv_in_kernel<<<n_max < 2048 ? n_max : 2048, 128, 0, st>>>(side, w.conv0_w, w.conv0_b, b.act1,
                                                                     count_ptr, n_max);
        break;
    case NET_CONV2:
        conv_mid_kernel<7, 8, CONV2_NB><<<cdiv(n_max, CONV2_NB), 256, CONV2_LDS, st>>>(
            b.act1, w.conv2_w, w.conv2_b, b.act2, count_ptr, n_max);
        break;
    case NET_CONV4:
        conv_mid_kernel<8, 9, CONV4_NB><<<cdiv(n_max, CONV4_NB), 256, CONV4_LDS, st>>>(
            b.act2, w.conv4_w, w.conv4_b, b.act3, count_ptr, n_max);
        break;
    case NET_FC:
        fc_kernel<<<64 * cdiv(cdiv(n_max, FC_BM), 8), 256, 0, st>>>(b.act3, w.fc_w, w.fc_b, b.fc, count_ptr,
                                                                    n_max, b.max_rows);
        break;
    case NET_HEADS: {
        int blocks = cdiv(n_max, 4);
        heads_kernel<<<blocks < 1024 ? blocks : 1024, 256, 0, st>>>(heads_input_f32(w, b), legal,
                                                                    logp_out, value_out, prior_out,
                                                                    count_ptr, n_max);
        break;
    }
    default: break;
    }
}

}  // namespace nnb

// Half-precision-input variant of the policy-value network kernels (gfx950).
//
// Same layer structure as net_kernels.hip (Connect4Net.forward,
// nnbattle/agents/alphazero/network.py:43-50), but activations and the repacked
// conv / linear weights are stored as IEEE half and multiplied on
// v_mfma_f32_32x32x16_f16 with float32 accumulation (16x the f32 matrix rate).  Bias,
// ReLU, the heads, log-softmax and tanh stay in float32.  On the seed-0 random-init
// network the outputs differ from the float32 CPU forward by about 1e-5 absolute.
#include <hip/hip_runtime.h>
#include <math.h>
#include "c4_bits.h"
#include "engine_layout.h"
#include "kernels.h"
#include "heads_math.h"
#include "mfma_tiles.h"

namespace nnb {

using c4::u64;

__device__ __forceinline__ int batch_count16(const int32_t *count_ptr, int n_fixed) {
    return count_ptr ? *count_ptr : n_fixed;
}

// ---------------------------------------------------------------------------
// Weight repacking.  One MFMA consumes 16 input channels: lane (n, h) holds channels
// 16*ks + 8*h + {0..7} of output channel n, i.e. one 16-byte load.
// ---------------------------------------------------------------------------
// Conv2d weight f32 [n=128][c=128][kh][kw] -> half [tap][ks=c/16][h=(c%16)/8][n][c%8]
__global__ void repack_conv_f16_kernel(const float *__restrict__ w, _Float16 *__restrict__ wp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * 128 * 16) return;
    int j = i & 7, n = (i >> 3) & 127, h = (i >> 10) & 1, ks = (i >> 11) & 7, tap = i >> 14;
    int c = ks * 16 + h * 8 + j, kh = tap >> 2, kw = tap & 3;
    wp[i] = (_Float16)w[((n * 128 + c) * 4 + kh) * 4 + kw];
}

// Linear weight f32 [n=1024][c*90 + hw] -> half [ks=kk/16][h][n][kk%8], kk = hw*128 + c
__global__ void repack_fc_f16_kernel(const float *__restrict__ w, _Float16 *__restrict__ wp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)1024 * 11520) return;
    int j = i & 7, n = (i >> 3) & 1023, h = (i >> 13) & 1;
    int ks = (int)(i >> 14);
    int kk = ks * 16 + h * 8 + j, hw = kk >> 7, c = kk & 127;
    wp[i] = (_Float16)w[(size_t)n * 11520 + c * 90 + hw];
}

// ---------------------------------------------------------------------------
// conv 128->128, 4x4, pad 2, + bias + ReLU (network.py:31-34): implicit GEMM on
// v_mfma_f32_32x32x16_f16.  in/out: half, channels-last.
//
// LDS image.  A block stages NB boards as "positions" of 128 channels (256 bytes, padded
// to 272 so consecutive positions fall on different banks).  Every image row carries two
// extra all-zero positions after its IW data positions, and the rows of all boards follow
// each other, so a tap that steps off the left or right edge lands on zeros by plain
// address arithmetic; two leading zero positions serve the first row and one more zero
// position serves taps that step off the top or bottom.  With that layout the position a
// lane reads is (position of its output cell) + dh*(IW+2) + dw: the 16 lanes that one
// ds_read_b128 services together hold 16 consecutive output cells (see `sigma`), hence 16
// distinct positions within a span of 18, hence (almost always) 16 different banks.
//
// Work split.  Wave w owns output channels 32w..32w+31 for all MT tiles of 32 output cells.
// Per tap and 16-channel k-step a wave issues MT MFMAs; the A fragments come from LDS through
// a ring of ARING registers (the read for MFMA f+ARING is issued right after MFMA f), the
// weight fragments stream L2 -> VGPR BRING-1 steps ahead.  The result tile goes back
// through LDS so the global stores are full 16-byte, row-contiguous writes.
// ---------------------------------------------------------------------------
#ifdef NNB_CONV_BENCH_CLOCKS
__device__ unsigned long long g_dbg_clk[2];
#endif
constexpr int CONV_RS = 17;                      // LDS stride of one position in 16-byte chunks


// ABLATE (timing experiments only, 0 in the product): 1 skip staging, 2 reuse one A fragment,
// 4 reuse one weight fragment, 8 skip the output stores.
// NW = 32-channel tiles per wave.  2 halves the LDS reads per MFMA (a wave then owns 64 channels of
// half the row tiles); it measured equal to 1 here - the clock rose, the cycle count with it - so
// the product uses 1.
//
// FUSE_IN (first 128->128 layer only): the block's input is not read from memory but computed in
// place - leaf encode (AlphaZeroAgent.preprocess, agent_code.py:76-89) and the 2->128 convolution
// (network.py:29-30) straight from the side boards into the LDS image, as two k-steps of MFMAs
// whose A operand is built from the board bits.
template <int IH, int IW, int NB, int WAVES_PER_SIMD, int BRING, int ARING, bool FUSE_IN = false, int ABLATE = 0, int NW = 1>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void conv_mid_f16_kernel(const _Float16 *__restrict__ in,
                                                                           const _Float16 *__restrict__ wp,
                                                                           const float *__restrict__ bias,
                                                                           _Float16 *__restrict__ out,
                                                                           const int32_t *__restrict__ count_ptr,
                                                                           int n_fixed,
                                                                           const u64 *__restrict__ side = nullptr,
                                                                           const float *__restrict__ w_in = nullptr,
                                                                           const float *__restrict__ b_in = nullptr) {
    constexpr int OH = IH + 1, OW = IW + 1, IP = IH * IW, OP = OH * OW, W2 = IW + 2;
    constexpr int M = NB * OP, MT = (M + 31) / 32, ROWS = NB * IP;
    constexpr int LEAD = 2, ZERO = LEAD + NB * IH * W2;          // position indices
    constexpr int NZ = 2 + NB * IH * 2 + 1;                      // zero positions to fill
    // wave grid: 4/NW groups of NW*32 output channels x NW groups of MTW row tiles
    static_assert(NW == 1 || NW == 2, "a wave owns one or two 32-channel tiles");
    constexpr int MTW = (MT + NW - 1) / NW;                      // row tiles per wave (the last group may own a dummy)
    constexpr int STEPS = 8 * MTW;                               // A fragments per tap
    static_assert(STEPS % ARING == 0, "ring depth must divide the A fragments per tap");
    constexpr int PB = CONV_RS * 16;                             // bytes per position
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);

    const int count = batch_count16(count_ptr, n_fixed);
    const int b0 = blockIdx.x * NB;
    if (b0 >= count) return;
    const int nb_valid = min(NB, count - b0);
    const int tid = threadIdx.x;

#ifdef NNB_CONV_BENCH_CLOCKS                                       // timing harness only: shader clock under this kernel's load
    const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_w0 = wall_clock64();
#endif
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    if constexpr (FUSE_IN) {
        static_assert(IH == 7 && IW == 8, "the fused input layer produces 7x8 maps");
        const int lane_ = tid & 63, w_ = tid >> 6, m_ = lane_ & 31, h_ = lane_ >> 5;
        // weights of this wave's 32 output channels: k = plane*16 + kh*4 + kw, lane half h_ holds k%16 in [8h_, 8h_+8)
        half8 bw[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bw[pl][jj] = (_Float16)w_in[(w_ * 32 + m_) * 32 + 16 * pl + 8 * h_ + jj];
        const float bv_in = b_in[w_ * 32 + m_];
        _Float16 *ldsh_in = reinterpret_cast<_Float16 *>(smem_raw);
        for (int j = 0; j < NB; ++j) {
            const bool have = j < nb_valid;
            const u64 planes[2] = {have ? side[2 * (b0 + j)] : 0, have ? side[2 * (b0 + j) + 1] : 0};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int pos = mt * 32 + m_, oh = pos >> 3, ow = pos & 7;     // output cell of this MFMA row
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    half8 a;
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int ih = oh + 2 * h_ + rr - 2;                   // kh = 2h_ + rr
                        // bits iw = 0..6 of reference row ih: stones at bit iw*7 + (5 - ih)
                        u64 t = (unsigned)ih < 6u ? (planes[pl] >> (5 - ih)) & c4::kBottomMask : 0;
                        unsigned row = (unsigned)((t * 0x41041041040ULL) >> 42) & 0x7F;
                        unsigned nib = ((row << 2) >> ow) & 0xF;               // bit kw <-> iw = ow + kw - 2
#pragma unroll
                        for (int kw = 0; kw < 4; ++kw) a[rr * 4 + kw] = ((nib >> kw) & 1) ? (_Float16)1.0f : (_Float16)0.0f;
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bw[pl], c, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h_;
                    if (p < IP)
                        ldsh_in[(LEAD + (j * IH + p / IW) * W2 + p % IW) * (PB / 2) + w_ * 32 + m_] =
                            (_Float16)fmaxf(c[r] + bv_in, 0.f);
                }
            }
        }
    } else {
        const uint4 *src = reinterpret_cast<const uint4 *>(in + (size_t)b0 * IP * 128);
        if (!(ABLATE & 1))
            for (int e = tid; e < ROWS * 16; e += 256) {
                int p = e >> 4, q = e & 15;
                uint4 v = (p < nb_valid * IP) ? src[e] : zero4;
                lds[(LEAD + (p / IW) * W2 + p % IW) * CONV_RS + q] = v;
            }
    }
    for (int e = tid; e < NZ * 16; e += 256) {
        int z = e >> 4, q = e & 15;
        int pos = z < 2 ? z : (z == NZ - 1 ? ZERO : LEAD + ((z - 2) >> 1) * W2 + IW + ((z - 2) & 1));
        lds[pos * CONV_RS + q] = zero4;
    }
    __syncthreads();

    const int lane = tid & 63, w = tid >> 6, m = lane & 31, h = lane >> 5;
    const int cg = NW == 1 ? w : (w & 1), rg = NW == 1 ? 0 : (w >> 1);     // channel group, row group
    // per tile: this lane's output cell -> oh (100 = no such cell) and the position it reads at tap (0,0)
    int geo[MTW];
    {
        const int cell = sigma(m);
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            int r = (rg * MTW + t) * 32 + cell;
            int j = r / OP, o = r % OP;
            int oh = (r < M) ? o / OW : 100;
            geo[t] = (oh << 16) | (LEAD + (j * IH + o / OW) * W2 + o % OW);
        }
    }
    auto row_addr = [&](int t, int tap) -> unsigned {           // LDS byte address for lane half h
        const int dh = (tap >> 2) - 2, dw = (tap & 3) - 2;
        int oh = geo[t] >> 16, p0 = geo[t] & 0xFFFF;
        bool ok = (unsigned)(oh + dh) < (unsigned)IH;
        int pos = ok ? p0 + dh * W2 + dw : ZERO;
        return (unsigned)(pos * PB + h * 16);
    };

    f32x16 acc[MTW][NW];
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int q = 0; q < NW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][q][r] = 0.f;

    const uint4 *wlane = reinterpret_cast<const uint4 *>(wp) + h * 128 + cg * (32 * NW) + m;
    uint4 bring[BRING][NW];
#pragma unroll
    for (int i = 0; i < BRING - 1; ++i)
#pragma unroll
        for (int q = 0; q < NW; ++q) bring[i][q] = wlane[i * 256 + q * 32];
    unsigned base[MTW], nbase[MTW];
    uint4 afrag[ARING];
#pragma unroll
    for (int t = 0; t < MTW; ++t) base[t] = row_addr(t, 0);
#pragma unroll
    for (int f = 0; f < ARING; ++f)
        afrag[f] = *reinterpret_cast<const uint4 *>(smem_raw + base[f % MTW] + (f / MTW) * 32);

    for (int tap = 0; tap < 16; ++tap) {
        const int ntap = tap < 15 ? tap + 1 : 15;
#pragma unroll
        for (int t = 0; t < MTW; ++t) nbase[t] = row_addr(t, ntap);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (!(ABLATE & 4)) {                                 // weights BRING-1 steps ahead
                int nxt = tap * 8 + ks + BRING - 1;
                nxt = nxt < 128 ? nxt : 127;
#pragma unroll
                for (int q = 0; q < NW; ++q) bring[(ks + BRING - 1) & (BRING - 1)][q] = wlane[nxt * 256 + q * 32];
            }
            half8 b[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) b[q] = as_half8(bring[(ABLATE & 4) ? 0 : (ks & (BRING - 1))][q]);
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
                const int f = ks * MTW + t, g = f + ARING;
                const half8 a = as_half8(afrag[(ABLATE & 2) ? 0 : f % ARING]);
#pragma unroll
                for (int q = 0; q < NW; ++q) acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[q], acc[t][q], 0, 0, 0);
                if (!(ABLATE & 2)) {
                    // refill the ring slot just consumed with the fragment of step f + ARING
                    if (g < STEPS)
                        afrag[g % ARING] = *reinterpret_cast<const uint4 *>(smem_raw + base[g % MTW] + (g / MTW) * 32);
                    else
                        afrag[g % ARING] = *reinterpret_cast<const uint4 *>(smem_raw + nbase[(g - STEPS) % MTW] +
                                                                            ((g - STEPS) / MTW) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);               // keep the MFMA / LDS-read interleave as written
            }
        }
#pragma unroll
        for (int t = 0; t < MTW; ++t) base[t] = nbase[t];
    }

    // epilogue: bias + ReLU, tile -> LDS as [output cell][128 channels] halves, then full-row stores
    __syncthreads();
    if (!(ABLATE & 8)) {
        _Float16 *ldsh = reinterpret_cast<_Float16 *>(smem_raw);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int ch = cg * (32 * NW) + q * 32 + m;
            const float bv = bias[ch];
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ic = (r & 3) + 8 * (r >> 2);              // MFMA row for lane half 0; +4 for half 1
                    const int R = (rg * MTW + t) * 32 + (h ? sigma(ic + 4) : sigma(ic));
                    if (R < M) ldsh[R * (PB / 2) + ch] = (_Float16)fmaxf(acc[t][q][r] + bv, 0.f);
                }
            }
        }
        __syncthreads();
        uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)b0 * OP * 128);
        const int nchunks = nb_valid * OP * 16;
        for (int e = tid; e < nchunks; e += 256) dst[e] = lds[(e >> 4) * CONV_RS + (e & 15)];
#ifdef NNB_CONV_BENCH_CLOCKS
        if (blockIdx.x == 700 && tid == 0) { g_dbg_clk[0] = __builtin_readcyclecounter() - dbg_c0; g_dbg_clk[1] = wall_clock64() - dbg_w0; }
#endif
    } else {
#ifdef NNB_CONV_BENCH_CLOCKS
        if (blockIdx.x == 700 && tid == 0) { g_dbg_clk[0] = __builtin_readcyclecounter() - dbg_c0; g_dbg_clk[1] = wall_clock64() - dbg_w0; }
#endif
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
            for (int q = 0; q < NW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc[t][q][r] == 12345.678f) out[0] = (_Float16)1.0f;   // keeps the accumulators live
    }
}

// ---------------------------------------------------------------------------
// conv 128->128, 4x4, pad 2, + bias + ReLU, "row tile" form.  Same arithmetic and the same
// kind of LDS image as conv_mid_f16_kernel, but an MFMA tile is ONE OUTPUT ROW of several
// boards (cells = board-major, NB*OW of them, padded to 32*MH), not 32 consecutive cells.
// For kernel row kh the output rows oh with oh+kh-2 outside [0, IH) read nothing but padding,
// and with this tiling that is a property of the whole tile: those MFMAs are not issued
// (conv 8x9 -> 9x10: 30 of the 36 (oh, kh) pairs remain; 7x8 -> 8x9: 26 of 32).  A lane's LDS
// address is one base plus a compile-time offset per (oh, kh, kw, k-step).
//
// Image: position = LEAD + j*BS + ih*(IW+2) + iw for board j; BS = IH*(IW+2) rounded up so
// that BS == OW (mod 16), which makes the 16 cells one ds_read_b128 services together land
// on 16 different bank groups even when they span two boards; every position that holds no
// input pixel is zero.  Block = 256*MH threads: wave w owns output channels 32*(w&3)..+31 and
// cells 32*(w>>2)..+31 of every row.
// ---------------------------------------------------------------------------

template <int IH, int IW, int ARING, int BRING, int KH, bool LAST>
__device__ __forceinline__ void conv_rows_body(f32x16 (&acc)[IH + 1], uint4 (&afrag)[ARING], uint4 (&bring)[BRING],
                                               const unsigned char *smem, int lane_base, int kw,
                                               const uint4 *__restrict__ wlane) {
    constexpr int W2 = IW + 2, PB = CONV_RS * 16;
    constexpr int T0 = rows_t0(KH), NT = rows_t1(IH, KH) - T0, STEPS = 8 * NT;
    constexpr int KHN = LAST ? KH + 1 : KH;                      // row of the body that runs next
    constexpr bool HAS_NEXT = !(LAST && KH == 3);
    constexpr int T0N = rows_t0(KHN & 3), NTN = rows_t1(IH, KHN & 3) - T0N;
    static_assert(STEPS % ARING == 0, "ring depth must divide the A fragments per tap");
    const int vb = lane_base + ((KH - 2) * W2 + kw - 2) * PB;
    const int vbn = LAST ? lane_base + ((KHN - 2) * W2 - 2) * PB : vb + PB;
    const int step0 = (KH * 4 + kw) * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        {                                                        // weights BRING-1 steps ahead
            int nxt = step0 + ks + BRING - 1;
            nxt = nxt < 128 ? nxt : 127;
            bring[(ks + BRING - 1) & (BRING - 1)] = wlane[nxt * 256];
        }
        const half8 b = as_half8(bring[ks & (BRING - 1)]);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int f = ks * NT + i, g = f + ARING;
            acc[T0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(afrag[f % ARING]), b, acc[T0 + i], 0, 0, 0);
            if (g < STEPS) {
                afrag[g % ARING] = *reinterpret_cast<const uint4 *>(smem + vb + (T0 + g % NT) * (W2 * PB) + (g / NT) * 32);
            } else if (HAS_NEXT) {
                const int gg = g - STEPS;
                afrag[g % ARING] = *reinterpret_cast<const uint4 *>(smem + vbn + (T0N + gg % NTN) * (W2 * PB) + (gg / NTN) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);                   // keep the MFMA / LDS-read interleave as written
        }
    }
}

template <int IH, int IW, int NB, int MH, int BRING, int ARING, bool FUSE_IN = false>
__global__ __launch_bounds__(256 * MH, MH == 1 ? 2 : 1) void conv_rows_f16_kernel(const _Float16 *__restrict__ in,
                                                                                  const _Float16 *__restrict__ wp,
                                                                                  const float *__restrict__ bias,
                                                                                  _Float16 *__restrict__ out,
                                                                                  const int32_t *__restrict__ count_ptr,
                                                                                  int n_fixed,
                                                                                  const u64 *__restrict__ side = nullptr,
                                                                                  const float *__restrict__ w_in = nullptr,
                                                                                  const float *__restrict__ b_in = nullptr) {
    constexpr int OH = IH + 1, OW = IW + 1, IP = IH * IW, OP = OH * OW, W2 = IW + 2;
    constexpr int RC = NB * OW;                                  // cells of one output row over the block's boards
    static_assert(RC <= 32 * MH, "one output row of all boards must fit the wave tiles");
    constexpr int BS = conv_board_stride(IH, IW), LEAD = 2;
    constexpr int NPOS = LEAD + NB * BS + 2;                     // positions in the image
    constexpr int PB = CONV_RS * 16;                             // bytes per position
    constexpr int NT_ = 256 * MH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);

    const int count = batch_count16(count_ptr, n_fixed);
    const int b0 = blockIdx.x * NB;
    if (b0 >= count) return;
    const int nb_valid = min(NB, count - b0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, m = lane & 31, h = lane >> 5;
    const int nt = w & 3, mh = w >> 2;                           // channel tile, cell half
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    if constexpr (FUSE_IN) {
        static_assert(IH == 7 && IW == 8, "the fused input layer produces 7x8 maps");
        // zero everything, then the 2->128 layer writes the data positions
        for (int e = tid; e < NPOS * CONV_RS; e += NT_) lds[e] = zero4;
        __syncthreads();
        half8 bw[2];                                             // k = plane*16 + kh*4 + kw; lane half h holds k%16 in [8h, 8h+8)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bw[pl][jj] = (_Float16)w_in[(nt * 32 + m) * 32 + 16 * pl + 8 * h + jj];
        const float bv_in = b_in[nt * 32 + m];
        _Float16 *ldsh_in = reinterpret_cast<_Float16 *>(smem_raw);
        for (int j = mh; j < nb_valid; j += MH) {                // boards are shared out over the cell halves
            const u64 planes[2] = {side[2 * (b0 + j)], side[2 * (b0 + j) + 1]};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int pos = mt * 32 + m, oh = pos >> 3, ow = pos & 7;      // output cell of this MFMA row
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    half8 a;
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int ih = oh + 2 * h + rr - 2;                    // kh = 2h + rr
                        // bits iw = 0..6 of reference row ih: stones at bit iw*7 + (5 - ih)
                        u64 t = (unsigned)ih < 6u ? (planes[pl] >> (5 - ih)) & c4::kBottomMask : 0;
                        unsigned row = (unsigned)((t * 0x41041041040ULL) >> 42) & 0x7F;
                        unsigned nib = ((row << 2) >> ow) & 0xF;               // bit kw <-> iw = ow + kw - 2
#pragma unroll
                        for (int kw = 0; kw < 4; ++kw) a[rr * 4 + kw] = ((nib >> kw) & 1) ? (_Float16)1.0f : (_Float16)0.0f;
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bw[pl], c, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (p < IP)
                        ldsh_in[(LEAD + j * BS + (p / IW) * W2 + p % IW) * (PB / 2) + nt * 32 + m] =
                            (_Float16)fmaxf(c[r] + bv_in, 0.f);
                }
            }
        }
    } else {
        // one pass over the image: input pixels from memory, zeros everywhere else
        const uint4 *src = reinterpret_cast<const uint4 *>(in + (size_t)b0 * IP * 128);
        for (int e = tid; e < NPOS * 16; e += NT_) {
            const int pos = e >> 4, q = e & 15;
            const int rel = pos - LEAD, j = rel / BS, rem = rel - j * BS, ih = rem / W2, iw = rem - ih * W2;
            const bool data = rel >= 0 && j < nb_valid && ih < IH && iw < IW;
            lds[pos * CONV_RS + q] = data ? src[((j * IH + ih) * IW + iw) * 16 + q] : zero4;
        }
    }
    __syncthreads();

    // this lane's cell of every output row: board j, column ow (lanes past the last cell repeat it)
    int cell = 32 * mh + sigma(m);
    cell = cell < RC ? cell : RC - 1;
    const int lane_base = (LEAD + (cell / OW) * BS + cell % OW) * PB + h * 16;

    f32x16 acc[OH];
#pragma unroll
    for (int t = 0; t < OH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const uint4 *wlane = reinterpret_cast<const uint4 *>(wp) + h * 128 + nt * 32 + m;
    uint4 bring[BRING];
#pragma unroll
    for (int i = 0; i < BRING - 1; ++i) bring[i] = wlane[i * 256];
    uint4 afrag[ARING];
    {
        constexpr int T0 = rows_t0(0), NT0 = rows_t1(IH, 0) - T0;
        const int vb = lane_base + (-2 * W2 - 2) * PB;
#pragma unroll
        for (int f = 0; f < ARING; ++f)
            afrag[f] = *reinterpret_cast<const uint4 *>(smem_raw + vb + (T0 + f % NT0) * (W2 * PB) + (f / NT0) * 32);
    }
#define NNB_ROWS_KH(KH)                                                                                       \
    {                                                                                                         \
        _Pragma("unroll 1") for (int kw = 0; kw < 3; ++kw)                                                    \
            conv_rows_body<IH, IW, ARING, BRING, KH, false>(acc, afrag, bring, smem_raw, lane_base, kw, wlane); \
        conv_rows_body<IH, IW, ARING, BRING, KH, true>(acc, afrag, bring, smem_raw, lane_base, 3, wlane);    \
    }
    NNB_ROWS_KH(0) NNB_ROWS_KH(1) NNB_ROWS_KH(2) NNB_ROWS_KH(3)
#undef NNB_ROWS_KH

    // epilogue: bias + ReLU, tile -> LDS as [board][output cell][128 channels] halves, then full-row stores
    __syncthreads();
    {
        const float bv = bias[nt * 32 + m];
        _Float16 *ldsh = reinterpret_cast<_Float16 *>(smem_raw);
#pragma unroll
        for (int t = 0; t < OH; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ic = (r & 3) + 8 * (r >> 2);                  // MFMA row for lane half 0; +4 for half 1
                const int c = 32 * mh + (h ? sigma(ic + 4) : sigma(ic));
                if (c < RC) {
                    const int R = (c / OW) * OP + t * OW + c % OW;
                    ldsh[R * (PB / 2) + nt * 32 + m] = (_Float16)fmaxf(acc[t][r] + bv, 0.f);
                }
            }
        }
        __syncthreads();
        uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)b0 * OP * 128);
        const int nchunks = nb_valid * OP * 16;
        for (int e = tid; e < nchunks; e += NT_) dst[e] = lds[(e >> 4) * CONV_RS + (e & 15)];
    }
}

constexpr size_t conv_rows_lds_bytes(int ih, int iw, int nb) {
    int npos = 2 + nb * conv_board_stride(ih, iw) + 2, m = nb * (ih + 1) * (iw + 1);
    return (size_t)(npos > m ? npos : m) * CONV_RS * 16;
}

// ---------------------------------------------------------------------------
// linear 11520->1024 (network.py:36-39,46-47) on v_mfma_f32_32x32x16_f16, split in two
// halves of K.  Block tile 128 rows x 256 columns, 4 waves (one per SIMD); wave w owns
// columns 64w..64w+63 (two 32-column tiles) for all four 32-row tiles, so each A fragment
// read from LDS feeds two MFMAs.  Raw float32 partial sums go to part[ksplit][row][1024];
// the heads kernel adds the halves, the bias and the ReLU.
//
// Pipeline (one wave per SIMD, so every latency is hidden in-wave): A slabs of 64 k go
// global -> registers four slabs ahead, registers -> LDS two slabs ahead (three LDS
// buffers, one barrier per slab); A fragments LDS -> registers one k-step ahead; weight
// fragments L2 -> registers seven k-steps ahead.
// Block order: the 32 blocks that share blockIdx % 8 (observed to share an XCD and its L2)
// cover 4 row blocks x (4 column blocks x 2 K halves), so each A slab is fetched into that
// L2 once and re-read by 8 blocks.
// ---------------------------------------------------------------------------
constexpr int FCH_K = 11520, FCH_N = 1024, FCH_BM = 128, FCH_BN = 256, FCH_BK = 64, FCH_SPLIT = 2;

// MTR = 32-row tiles per block (1..4): the kernel picks the smallest that covers the batch with the
// launched blocks, so a partly filled batch costs proportionally less instead of leaving CUs idle.
// WAVES = waves per block: 4 (a block owns 256 columns) or 1 (64 columns per block, MTR = 1 only: the
// small-batch form, 16 column blocks x 2 K halves per row block so that a handful of rows still spreads
// the 23.6 MB of weights over many CUs).  A wave's arithmetic is the same either way.
template <int MTR, int ABLATE, int BR, int NSET, int WAVES = 4>
__device__ __forceinline__ void fc_f16_tile(const _Float16 *__restrict__ a, const _Float16 *__restrict__ wp,
                                            float *__restrict__ part, int count, int max_rows,
                                            uint4 (&lds)[3][FCH_BM * 8]) {
    static_assert(WAVES == 4 || (WAVES == 1 && MTR == 1), "one-wave blocks are the small-batch form");
    const int bid = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, h = lane >> 5;
    int mb, nb, ksp, w;
    if constexpr (WAVES == 4) {
        const int x = bid & 7, i = bid >> 3;                // i in [0, 32 * groups)
        const int grp = i >> 5, ii = i & 31;
        mb = (grp * 8 + x) * 4 + (ii & 3);
        nb = (ii >> 2) & 3; ksp = ii >> 4;
        w = tid >> 6;
    } else {
        const int q = bid & 31;
        mb = bid >> 5; nb = q & 3; w = (q >> 2) & 3; ksp = q >> 4;
    }
    const int row0 = mb * (32 * MTR);
    if (row0 >= count) return;
#ifdef NNB_CONV_BENCH_CLOCKS
    const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_w0 = wall_clock64();
#endif

    constexpr int KHALF = FCH_K / FCH_SPLIT;                // 5760
    constexpr int NSLAB = KHALF / FCH_BK;                   // 90 slabs of 64 k = 4 k-steps each
    constexpr int NKS = NSLAB * 4;
    static_assert(24 % BR == 0 && 6 % NSET == 0, "ring depths must divide the 6-slab loop body");
    const int k0 = ksp * KHALF;

    // staging map: 128 rows x 8 chunks of 16 bytes = 1024 chunks, 4 per thread.  Kept in named
    // scalars (not arrays): arrays of staging registers ended up in scratch memory.
#define FC_MAP(C)                                                                                 \
    int sdst##C;                                                                                  \
    const uint4 *abase##C;                                                                        \
    {                                                                                             \
        int e = tid + 64 * WAVES * (C), srow = e >> 3, scol = e & 7;                              \
        sdst##C = srow * 8 + (scol ^ ((srow >> 1) & 7));                                          \
        int r = min(row0 + srow, max_rows - 1);                                                   \
        abase##C = reinterpret_cast<const uint4 *>(a + (size_t)r * FCH_K + k0) + scol;            \
    }
    FC_MAP(0) FC_MAP(1) FC_MAP(2) FC_MAP(3)     // chunk C belongs to row tile C; tiles >= MTR are never touched
#undef FC_MAP
    constexpr int NCHUNK = MTR * 4 / WAVES;                 // 16-byte staging chunks per thread and slab
#define FC_LD1(S, C, SLAB) if constexpr ((C) < NCHUNK) S##C = abase##C[(SLAB) * 8];
#define FC_WR1(S, C, BUF) if constexpr ((C) < NCHUNK) lds[BUF][sdst##C] = S##C;
#define FC_LOAD(S, SLAB) { FC_LD1(S, 0, SLAB) FC_LD1(S, 1, SLAB) FC_LD1(S, 2, SLAB) FC_LD1(S, 3, SLAB) }
#define FC_WRITE(S, BUF) { FC_WR1(S, 0, BUF) FC_WR1(S, 1, BUF) FC_WR1(S, 2, BUF) FC_WR1(S, 3, BUF) }
    // A-fragment read offsets (16-byte units) inside a buffer, per row tile, for k-step 0 of a slab
    int aoff[4], asw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = t * 32 + m;
        aoff[t] = row * 8;
        asw[t] = (row >> 1) & 7;
    }

    // NSET register sets for the global -> LDS staging (named scalars, see above): a slab is loaded
    // NSET slabs before it is written to LDS, and written two slabs before it is used.
    uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3, sc0, sc1, sc2, sc3, sd0, sd1, sd2, sd3, se0, se1, se2, se3, sf0, sf1, sf2, sf3;
    // prologue: slabs 0 and 1 into LDS, slabs 2 .. 1+NSET in flight in the register sets
    FC_LOAD(sa, 0) FC_LOAD(sb, 1)
    FC_WRITE(sa, 0) FC_WRITE(sb, 1)
    if constexpr (NSET == 2) { FC_LOAD(sa, 2) FC_LOAD(sb, 3) }
    if constexpr (NSET == 3) { FC_LOAD(sa, 2) FC_LOAD(sb, 3) FC_LOAD(sc, 4) }
    if constexpr (NSET == 6) { FC_LOAD(sa, 2) FC_LOAD(sb, 3) FC_LOAD(sc, 4) FC_LOAD(sd, 5) FC_LOAD(se, 6) FC_LOAD(sf, 7) }
    __syncthreads();

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // weight fragment of (k-step, column tile): ((kstep*2 + h)*1024 + n) * 16 bytes
    const uint4 *wlane = reinterpret_cast<const uint4 *>(wp) + (size_t)(k0 / 16) * 2 * FCH_N + h * FCH_N +
                         nb * FCH_BN + w * 64 + m;
    uint4 br0[BR], br1[BR];                                 // ring of weight fragments, column tiles 0 / 1
#pragma unroll
    for (int j = 0; j < BR - 1; ++j) { br0[j] = wlane[(size_t)j * 2 * FCH_N]; br1[j] = wlane[(size_t)j * 2 * FCH_N + 32]; }
    uint4 afa[4], afb[4];                                   // A fragments of even / odd k-steps
#pragma unroll
    for (int t = 0; t < MTR; ++t) afa[t] = lds[0][aoff[t] + (h ^ asw[t])];

// one k-step: J = k-step index inside the 24-step body, KS = k-step inside its slab,
// CUR / NXT = fragment sets (afa, afb), NBUF = LDS buffer the next step's fragments come from.
// Each of the 8 MFMAs is followed by ONE memory instruction - two weight loads, the four
// fragment reads of the next k-step, then two slots X0 / X1 that the slab macro fills with
// staging work - and a scheduling fence, so the MFMA pipe never waits behind a run of loads.
#define FC_MM(T, U_, CUR) if constexpr ((T) < MTR) acc[T][U_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(((ABLATE & 2) ? afa : CUR)[T]), U_ ? b1 : b0, acc[T][U_], 0, 0, 0);
#define FC_RD(T, KS, NXT, NBUF)                                                                           \
    if constexpr ((T) < MTR) if (((KS) < 3 || s + 1 < NSLAB) && !(ABLATE & 2)) NXT[T] = lds[NBUF][aoff[T] + ((2 * (((KS) + 1) & 3) + h) ^ asw[T])];
#define FC_SB __builtin_amdgcn_sched_barrier(0);
#define FC_KSTEP(J, KS, CUR, NXT, NBUF, X0, X1)                                                           \
    {                                                                                                     \
        const half8 b0 = as_half8(br0[(ABLATE & 4) ? 0 : (J) % BR]), b1 = as_half8(br1[(ABLATE & 4) ? 0 : (J) % BR]); \
        int nxt = s * 4 + (KS) + BR - 1;                                                                  \
        nxt = nxt < NKS ? nxt : NKS - 1;                                                                  \
        FC_MM(0, 0, CUR) if (!(ABLATE & 4)) br0[((J) + BR - 1) % BR] = wlane[(size_t)nxt * 2 * FCH_N]; FC_SB      \
        FC_MM(0, 1, CUR) if (!(ABLATE & 4)) br1[((J) + BR - 1) % BR] = wlane[(size_t)nxt * 2 * FCH_N + 32]; FC_SB \
        FC_MM(1, 0, CUR) FC_RD(0, KS, NXT, NBUF) FC_SB                                                    \
        FC_MM(1, 1, CUR) FC_RD(1, KS, NXT, NBUF) FC_SB                                                    \
        FC_MM(2, 0, CUR) FC_RD(2, KS, NXT, NBUF) FC_SB                                                    \
        FC_MM(2, 1, CUR) FC_RD(3, KS, NXT, NBUF) FC_SB                                                    \
        FC_MM(3, 0, CUR) X0 FC_SB                                                                         \
        FC_MM(3, 1, CUR) X1 FC_SB                                                                         \
    }

// one slab: U = slab index inside the 6-slab body (LDS buffer U % 3), STG = staging register set.
// Slab s+2 goes registers -> LDS during k-steps 0 and 1, slab s+2+NSET global -> registers during 2 and 3.
#define FC_W2(STG, BUF, A, B) if (s + 2 < NSLAB && !(ABLATE & 16)) { FC_WR1(STG, A, BUF) FC_WR1(STG, B, BUF) }
#define FC_L2(STG, A, B)                                                                                  \
    if (s + 2 + NSET < NSLAB && !(ABLATE & 1)) { FC_LD1(STG, A, s + 2 + NSET) FC_LD1(STG, B, s + 2 + NSET) }
#define FC_SLAB(U, STG)                                                                                   \
    {                                                                                                     \
        const int s = s6 + (U);                                                                           \
        FC_KSTEP((U) * 4 + 0, 0, afa, afb, (U) % 3, FC_W2(STG, ((U) + 2) % 3, 0, 1), FC_W2(STG, ((U) + 2) % 3, 2, 3)) \
        FC_KSTEP((U) * 4 + 1, 1, afb, afa, (U) % 3, , )                                                   \
        FC_KSTEP((U) * 4 + 2, 2, afa, afb, (U) % 3, FC_L2(STG, 0, 1), FC_L2(STG, 2, 3))                   \
        FC_KSTEP((U) * 4 + 3, 3, afb, afa, ((U) + 1) % 3, , )                                             \
        if (!(ABLATE & 8)) __syncthreads();                                                               \
    }

    for (int s6 = 0; s6 < NSLAB; s6 += 6) {
        if constexpr (NSET == 2) { FC_SLAB(0, sa) FC_SLAB(1, sb) FC_SLAB(2, sa) FC_SLAB(3, sb) FC_SLAB(4, sa) FC_SLAB(5, sb) }
        if constexpr (NSET == 3) { FC_SLAB(0, sa) FC_SLAB(1, sb) FC_SLAB(2, sc) FC_SLAB(3, sa) FC_SLAB(4, sb) FC_SLAB(5, sc) }
        if constexpr (NSET == 6) { FC_SLAB(0, sa) FC_SLAB(1, sb) FC_SLAB(2, sc) FC_SLAB(3, sd) FC_SLAB(4, se) FC_SLAB(5, sf) }
    }
#ifdef NNB_CONV_BENCH_CLOCKS
    if (blockIdx.x == 100 && tid == 0) { g_dbg_clk[0] = __builtin_readcyclecounter() - dbg_c0; g_dbg_clk[1] = wall_clock64() - dbg_w0; }
#endif
#undef FC_SLAB
#undef FC_KSTEP
#undef FC_MM
#undef FC_RD
#undef FC_SB
#undef FC_W2
#undef FC_L2
#undef FC_LOAD
#undef FC_WRITE
#undef FC_LD1
#undef FC_WR1

    float *obase = part + (size_t)ksp * max_rows * FCH_N + nb * FCH_BN + w * 64 + m;
#pragma unroll
    for (int t = 0; t < MTR; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < count) {
                obase[(size_t)row * FCH_N] = acc[t][0][r];
                obase[(size_t)row * FCH_N + 32] = acc[t][1][r];
            }
        }
    }
}

// Small batches: one wave per block (see WAVES).  Launch with 32 blocks of 64 threads per 32 rows.
__global__ __launch_bounds__(64) void fc_f16_small_kernel(const _Float16 *__restrict__ a, const _Float16 *__restrict__ wp,
                                                          float *__restrict__ part,
                                                          const int32_t *__restrict__ count_ptr, int n_fixed,
                                                          int max_rows) {
    __shared__ __attribute__((aligned(16))) uint4 lds[3][FCH_BM * 8];
    fc_f16_tile<1, 0, 8, 2, 1>(a, wp, part, batch_count16(count_ptr, n_fixed), max_rows, lds);
}

// ABLATE (timing experiments only): 1 no A global loads, 2 no A-fragment LDS reads, 4 no weight loads,
// 8 no barriers, 16 no staging writes.
template <int ABLATE = 0, int BR = 8, int NSET = 2>
__global__ __launch_bounds__(256, 1) void fc_f16_kernel(const _Float16 *__restrict__ a,   // [rows][11520]
                                                        const _Float16 *__restrict__ wp,  // repacked
                                                        float *__restrict__ part,         // [2][max_rows][1024]
                                                        const int32_t *__restrict__ count_ptr, int n_fixed,
                                                        int max_rows) {
    __shared__ __attribute__((aligned(16))) uint4 lds[3][FCH_BM * 8];      // 128 rows x 64 halfs, three buffers
    const int count = batch_count16(count_ptr, n_fixed);
    const int mblocks = gridDim.x >> 3;                     // row blocks the grid provides (8 column/K blocks each)
    const int tiles = (count + 32 * mblocks - 1) / (32 * mblocks);
    if (tiles <= 1) fc_f16_tile<1, ABLATE, BR, NSET>(a, wp, part, count, max_rows, lds);
    else if (tiles == 2) fc_f16_tile<2, ABLATE, BR, NSET>(a, wp, part, count, max_rows, lds);
    else if (tiles == 3) fc_f16_tile<3, ABLATE, BR, NSET>(a, wp, part, count, max_rows, lds);
    else fc_f16_tile<4, ABLATE, BR, NSET>(a, wp, part, count, max_rows, lds);
}

// ---------------------------------------------------------------------------
// Heads over split linear-layer partial sums: x = ReLU(part[0] + part[1] + bias), then
// 1024->7 log-softmax, 1024->1 tanh and the legal-move mask (heads_math.h).  One wave per row;
// the 8 head rows stay in registers for all rows of a wave.  The search itself does not launch
// this kernel (the tree step finishes its own leaf); it serves nnb_net_forward and the checks.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heads_split_kernel(HeadsDev hd, const uint8_t *__restrict__ legal,
                                                          float *__restrict__ logp_out, float *__restrict__ value_out,
                                                          float *__restrict__ prior_out,
                                                          const int32_t *__restrict__ count_ptr, int n_fixed) {
    const int count = batch_count16(count_ptr, n_fixed);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    if (wave >= count) return;
    float4 wv[8][4];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[o][i] = reinterpret_cast<const float4 *>(hd.hw + o * 1024)[lane + 64 * i];
    for (int row = wave; row < count; row += nwaves) {
        float4 xv[4];
        heads_load_row(hd, row, lane, xv);
        float z[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s = __builtin_fmaf(xv[i].x, wv[o][i].x, s);
                s = __builtin_fmaf(xv[i].y, wv[o][i].y, s);
                s = __builtin_fmaf(xv[i].z, wv[o][i].z, s);
                s = __builtin_fmaf(xv[i].w, wv[o][i].w, s);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            z[o] = s + hd.hb[o];
        }
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
static inline int cdiv16(int a, int b) { return (a + b - 1) / b; }

HeadsDev heads_input_f16(const NetWeights &w, const NetBuffers &b) {
    return HeadsDev{nullptr, b.fcpart, b.fcpart + (size_t)b.max_rows * 1024, w.fc_b, w.heads_w, w.heads_b, b.max_rows};
}

constexpr int C2H_NB = 4, C4H_NB = 3;
constexpr size_t conv_lds_bytes(int ih, int iw, int nb) {
    int npos = 2 + nb * ih * (iw + 2) + 1, m = nb * (ih + 1) * (iw + 1);
    return (size_t)(npos > m ? npos : m) * CONV_RS * 16;
}
// Engines of at most kSmallBatch slots (single searches, small tournaments) are latency-bound: one board
// per block gives every wave a third of the tiles and three to four times as many blocks.
constexpr int kSmallBatch = 128;
constexpr size_t C2S_LDS = conv_lds_bytes(7, 8, 1), C4S_LDS = conv_lds_bytes(8, 9, 1);
constexpr size_t C2H_LDS = conv_lds_bytes(7, 8, C2H_NB);
constexpr size_t C4H_LDS = conv_rows_lds_bytes(8, 9, C4H_NB);

hipError_t prepare_net_kernels_f16() {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mid_f16_kernel<7, 8, C2H_NB, 2, 4, 4, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2H_LDS);
    if (err != hipSuccess) return err;
    err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mid_f16_kernel<7, 8, 1, 2, 8, 4, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2S_LDS);
    if (err != hipSuccess) return err;
    err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_mid_f16_kernel<8, 9, 1, 2, 8, 4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4S_LDS);
    if (err != hipSuccess) return err;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rows_f16_kernel<8, 9, C4H_NB, 1, 4, 8>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4H_LDS);
}

void launch_repack_f16(hipStream_t st, const NetWeights &dst, const float *conv2_w, const float *conv4_w,
                       const float *fc_w) {
    repack_conv_f16_kernel<<<cdiv16(128 * 128 * 16, 256), 256, 0, st>>>(conv2_w, (_Float16 *)dst.conv2_wh);
    repack_conv_f16_kernel<<<cdiv16(128 * 128 * 16, 256), 256, 0, st>>>(conv4_w, (_Float16 *)dst.conv4_wh);
    repack_fc_f16_kernel<<<(1024 * 11520) / 256, 256, 0, st>>>(fc_w, (_Float16 *)dst.fc_wh);
}

void launch_net_stage_f16(hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
                          const uint8_t *legal, float *logp_out, float *value_out, float *prior_out,
                          const int32_t *count_ptr, int n_max) {
    if (n_max <= 0) return;
    _Float16 *a2 = reinterpret_cast<_Float16 *>(b.act2), *a3 = reinterpret_cast<_Float16 *>(b.act3);
    switch (stage) {
    case NET_CONV0:
        break;      // fused into the first 128->128 layer below
    case NET_CONV2:
        if (n_max <= kSmallBatch) {
            conv_mid_f16_kernel<7, 8, 1, 2, 8, 4, true><<<n_max, 256, C2S_LDS, st>>>(
                nullptr, (const _Float16 *)w.conv2_wh, w.conv2_b, a2, count_ptr, n_max, side, w.conv0_w, w.conv0_b);
            break;
        }
        conv_mid_f16_kernel<7, 8, C2H_NB, 2, 4, 4, true><<<cdiv16(n_max, C2H_NB), 256, C2H_LDS, st>>>(
            nullptr, (const _Float16 *)w.conv2_wh, w.conv2_b, a2, count_ptr, n_max, side, w.conv0_w, w.conv0_b);
        break;
    case NET_CONV4:
        if (n_max <= kSmallBatch) {
            conv_mid_f16_kernel<8, 9, 1, 2, 8, 4><<<n_max, 256, C4S_LDS, st>>>(
                a2, (const _Float16 *)w.conv4_wh, w.conv4_b, a3, count_ptr, n_max);
            break;
        }
        conv_rows_f16_kernel<8, 9, C4H_NB, 1, 4, 8><<<cdiv16(n_max, C4H_NB), 256, C4H_LDS, st>>>(
            a2, (const _Float16 *)w.conv4_wh, w.conv4_b, a3, count_ptr, n_max);
        break;
    case NET_FC: {
        if (n_max <= kSmallBatch) {
            fc_f16_small_kernel<<<32 * cdiv16(n_max, 32), 64, 0, st>>>(a3, (const _Float16 *)w.fc_wh, b.fcpart, count_ptr,
                                                                    n_max, b.max_rows);
            break;
        }
        int groups = cdiv16(cdiv16(n_max, FCH_BM), 32);      // 32 row blocks per group of 256 blocks
        fc_f16_kernel<0, 8, 2><<<256 * groups, 256, 0, st>>>(a3, (const _Float16 *)w.fc_wh, b.fcpart, count_ptr, n_max,
                                                    b.max_rows);
        break;
    }
    case NET_HEADS: {
        int blocks = cdiv16(n_max, 4);                          // one wave per row, at most 2 blocks per CU
        heads_split_kernel<<<blocks < 512 ? blocks : 512, 256, 0, st>>>(heads_input_f16(w, b), legal, logp_out, value_out,
                                                                        prior_out, count_ptr, n_max);
        break;
    }
    default: break;
    }
}

}  // namespace nnb

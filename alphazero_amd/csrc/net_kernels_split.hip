// Float32-faithful variant of the policy-value network kernels (gfx950) on the half-input matrix cores.
//
// Connect4Net.forward (nnbattle/agents/alphazero/network.py:43-50) multiplies in float32.  The exact
// f32 MFMA (v_mfma_f32_32x32x2_f32, net_kernels.hip) runs at 1/16 of the half-input rate.  Here every
// float32 operand x is carried as a PAIR of halves, x*s = hi + lo with hi = half(x*s) and
// lo = half(x*s - hi), s a power of two that puts the largest magnitude of the tensor just under 2^14
// (so hi never overflows and lo stays a normal half for everything within 2^-17 of the maximum).
// hi + lo reproduces x*s to 2^-22 relative (float32 keeps 2^-24), and a product becomes three MFMAs
//     a*w*(sa*sw) ~= ah*wh + ah*wl + al*wh          (al*wl, 2^-22 of the product, is dropped)
// on v_mfma_f32_16x16x32_f16 (the 2->128 layer: 32x32x16) with float32 accumulation; the epilogue multiplies by
// 1/(sa*sw), exactly.
// Weights get one scale per layer at load time.  Activations get one scale PER BOARD, taken from that
// board's own maximum by the kernel that produces them, so a position's outputs do not depend on what
// else is in the batch.  Bias, ReLU, the heads, log-softmax and tanh are float32 as in the other paths.
//
// Activation layout in memory: [board][cell][256 halves] = the 128 hi halves of the cell's channels,
// then the 128 lo halves (512 bytes per cell), plus one float per board: 1/s.
#include <hip/hip_runtime.h>
#include <math.h>
#include "c4_bits.h"
#include "engine_layout.h"
#include "kernels.h"
#include "heads_math.h"
#include "mfma_tiles.h"

namespace nnb {

using c4::u64;

constexpr int SP_RS = 33;                 // 16-byte chunks per LDS position: 16 hi, 16 lo, one pad (33 == 1 mod 16)
constexpr int SP_PB = SP_RS * 16;         // bytes per position
constexpr int SP_LO = 256;                // byte offset of the lo halves inside a position

__device__ __forceinline__ int batch_count_sp(const int32_t *count_ptr, int n_fixed) {
    return count_ptr ? *count_ptr : n_fixed;
}

// Power of two s with m*s in [2^13, 2^14) for a finite m >= 2^-113, else 1; inv = 1/s.
__device__ __forceinline__ float pow2_scale(float m, float &inv) {
    const unsigned e = (__float_as_uint(m) >> 23) & 0xFFu;
    if (e < 14u || e == 255u) { inv = 1.0f; return 1.0f; }
    inv = __uint_as_float((e - 13u) << 23);
    return __uint_as_float((267u - e) << 23);
}

__device__ __forceinline__ void split_half(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);       // exact difference, then rounded to half
}

// ---------------------------------------------------------------------------
// Weight preparation (once per nnb_load_weights)
// ---------------------------------------------------------------------------
// max |w| over a tensor, as float bits (order-preserving for non-negative floats)
__global__ void absmax_kernel(const float *__restrict__ w, size_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// One thread: the layer scales from the maxima, and the static scale of the 2->128 layer's output.
// Its inputs are 0/1 planes, so an output of channel n is at most bias[n] + the sum of n's positive weights.
__global__ void split_scales_kernel(const unsigned *__restrict__ maxbits, const float *__restrict__ w0,
                                    const float *__restrict__ b0, SplitScales *__restrict__ sc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float inv;
    sc->sw0 = pow2_scale(__uint_as_float(maxbits[0]), inv); sc->inv_sw0 = inv;
    sc->sw2 = pow2_scale(__uint_as_float(maxbits[1]), inv); sc->inv_sw2 = inv;
    sc->sw4 = pow2_scale(__uint_as_float(maxbits[2]), inv); sc->inv_sw4 = inv;
    sc->swf = pow2_scale(__uint_as_float(maxbits[3]), inv); sc->inv_swf = inv;
    float bound = 0.f;
    for (int n = 0; n < 128; ++n) {
        float s = b0[n];
        for (int k = 0; k < 32; ++k) s += fmaxf(w0[n * 32 + k], 0.f);
        bound = fmaxf(bound, s);
    }
    sc->sa1 = pow2_scale(bound, inv); sc->inv_sa1 = inv;
}

// Conv2d weight f32 [n=128][c=128][kh][kw] -> hi / lo halves [tap][ks=c/16][h=(c%16)/8][n][c%8]
__global__ void repack_conv_split_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                         _Float16 *__restrict__ wh, _Float16 *__restrict__ wl) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * 128 * 16) return;
    int j = i & 7, n = (i >> 3) & 127, h = (i >> 10) & 1, ks = (i >> 11) & 7, tap = i >> 14;
    int c = ks * 16 + h * 8 + j, kh = tap >> 2, kw = tap & 3;
    split_half(w[((n * 128 + c) * 4 + kh) * 4 + kw] * *scale, wh[i], wl[i]);
}

// Linear weight f32 [n=1024][c*90 + hw] -> hi / lo halves [ks=kk/16][h][n][kk%8], kk = hw*128 + c
__global__ void repack_fc_split_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                       _Float16 *__restrict__ wh, _Float16 *__restrict__ wl) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)1024 * 11520) return;
    int j = i & 7, n = (i >> 3) & 1023, h = (i >> 13) & 1;
    int ks = (int)(i >> 14);
    int kk = ks * 16 + h * 8 + j, hw = kk >> 7, c = kk & 127;
    split_half(w[(size_t)n * 11520 + c * 90 + hw] * *scale, wh[i], wl[i]);
}

// 2->128 layer: [n][k = plane*16 + kh*4 + kw] keeps its order
__global__ void repack_conv0_split_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                          _Float16 *__restrict__ wh, _Float16 *__restrict__ wl) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 128 * 32) return;
    split_half(w[i] * *scale, wh[i], wl[i]);
}

// The hi or lo weights of one 128->128 layer (16 taps x 8 k-steps x 4 KB) as a buffer resource: a fragment load is
// then one instruction with a per-lane 32-bit offset and a scalar offset, and a read past the end returns zeros.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_weight_rsrc(const u32x4 *w) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(w), 0, 128 * 128 * 16 * 2, 0x00027000);
}

#ifdef NNB_SPLIT_BENCH_CLOCKS
// timing harness only (tools/split_bench.hip): shader-clock stamps of one block's phases
__device__ unsigned long long g_sp_clk[8];
__device__ unsigned long long g_sp_block_t[2 * 8192];         // wall clock (100 MHz) at the start and end of every block
#define NNB_SP_BLOCK 100
#define NNB_SP_STAMP(i) if (blockIdx.x == NNB_SP_BLOCK && threadIdx.x == 0) g_sp_clk[i] = __builtin_readcyclecounter();
#define NNB_SP_WALL(i) { if (blockIdx.x == NNB_SP_BLOCK && threadIdx.x == 0) g_sp_clk[i] = wall_clock64(); \
                         if (threadIdx.x == 0 && blockIdx.x < 8192) g_sp_block_t[2 * blockIdx.x + ((i) == 6)] = wall_clock64(); }
#else
#define NNB_SP_STAMP(i)
#define NNB_SP_WALL(i)
#endif

// LDS image of one board's input to a 4x4 / pad 2 layer: per pixel one position of 528 bytes (128 hi halves, 128 lo
// halves, pad), two lead positions, 12 image rows of iw + 2 positions (two rows of zeros above the board, the board's
// rows with two zero positions behind each, zeros below), two tail positions.  Every tap of every output cell is then
// an unconditional read: the lane's base position plus a compile-time offset, the lo fragment 256 bytes behind the hi.
constexpr int conv_board_positions(int ih, int iw) { return 2 + (ih + 4 > 12 ? ih + 4 : 12) * (iw + 2) + 2; }
constexpr size_t conv_board_lds_bytes() {
    int a = conv_board_positions(7, 8), b = conv_board_positions(8, 9), c = 9 * 10;
    int m = a > b ? a : b;
    return (size_t)(m > c ? m : c) * SP_PB + 64;                // + the per-wave board maxima
}

// ---------------------------------------------------------------------------
// All three convolutions, ONE board per block, 16-cell tiles (conv_board_split_kernel).
//
// The board's output cells are the rows of v_mfma_f32_16x16x32_f16 tiles, 16 at a time: 72 cells = 5 tiles for the
// 7x8 -> 8x9 layer, 90 cells = 6 tiles for 8x9 -> 9x10 (a 32-row tile of one board would waste a quarter of the last
// one; one output row of three boards fills 27 and 30 of 32 rows, but three boards' images take 145 KB of LDS: one
// block per CU, one wave per SIMD, nothing to run under the 18 % of a block's cycles that are not matrix work -
// measured 724 us against 637 us per 3979 boards).  Which cells share a tile is BoardTiles' business: cells on the
// same border share the taps that read only padding, and those (tile, tap) pairs are known at compile time and
// skipped.  The image of one board takes 72 KB of LDS and a wave at most 256 registers, so TWO blocks share a CU:
// while one is in its prologue or one of its epilogues (no matrix work), the other one's waves keep the matrix pipe
// busy - on a power-capped chip more pipes busy at a lower clock deliver more than fewer at a higher one.  The price
// is the weight stream: every block reads both layers' weights from L2 for one board instead of three.
// Wave w owns output channels 32w .. 32w+31 as two 16-column tiles.  Operand layouts of the instruction: lane l
// supplies row l % 16 of A and column l % 16 of B for k = 8 * (l / 16) .. + 7, and holds D[4 * (l / 16) + r][l % 16].
// One k-step is 32 input channels; the weight layout [tap][c / 16][(c % 16) / 8][n][c % 8] is the 32x32x16 one read
// two 16-channel steps at a time.
// ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int IH, int IW>
struct BoardGeom {
    static constexpr int OH = IH + 1, OW = IW + 1, NC = OH * OW, NT = (NC + 15) / 16, W2 = IW + 2, LEAD = 2 + 2 * W2;
    static constexpr int NPOS = conv_board_positions(IH, IW);
    static constexpr int row_first(int t) { return (16 * t) / OW; }
    static constexpr int row_last(int t) { return (16 * t + 15 < NC ? 16 * t + 15 : NC - 1) / OW; }
    // kernel row kh of tile t reads at least one input row
    static constexpr bool live(int t, int kh) { return row_last(t) + kh - 2 >= 0 && row_first(t) + kh - 2 < IH; }
};

// Which output cells share a 16-row tile in conv_board_split_kernel.  A tile pays for a tap as soon as ONE of its cells
// reads a pixel through it, so cells are grouped by the taps that fall on the zero border for all of them:
//   7x8 -> 8x9   tile 0 = output column 0 (8 cells: kernel columns 0 and 1 read only padding), tiles 1-4 = two output
//                rows x columns 1-8 (rows 0-1 skip kernel row 0, rows 6-7 kernel row 3): 64 (tile, tap) pairs of 80
//   8x9 -> 9x10  tile 0 = output row 0 (10 cells: kernel rows 0 and 1 read only padding), tiles 1-5 = two output
//                columns x rows 1-8 (columns 0-1 skip kernel column 0, columns 8-9 kernel column 3): 80 of 96
// against 68 and 84 with the cells taken row-major.  A skipped tap multiplies zeros, so every accumulator still sums
// the same products in the same order as with any other grouping (conv_spread_* keep the row-major one).
template <int IH, int IW>
struct BoardTiles {
    static constexpr int OH = IH + 1, OW = IW + 1, NC = OH * OW, W2 = IW + 2, LEAD = 2 + 2 * W2;
    static constexpr bool ROWS = OW == 9;                        // row-pair tiles (8 columns wide), else column-pair tiles
    static_assert(ROWS ? OH % 2 == 0 : (OH == 9 && OW % 2 == 0), "the two layers of this network");
    static constexpr int NT = 1 + (ROWS ? OH / 2 : OW / 2);
    static constexpr int cells(int t) { return t > 0 ? 16 : ROWS ? OH : OW; }
    // output row and column of cell i < cells(t) of tile t
    static constexpr int row(int t, int i) { return ROWS ? (t == 0 ? i : 2 * (t - 1) + (i >> 3)) : (t == 0 ? 0 : 1 + (i >> 1)); }
    static constexpr int col(int t, int i) { return ROWS ? (t == 0 ? 0 : 1 + (i & 7)) : (t == 0 ? i : 2 * (t - 1) + (i & 1)); }
    static constexpr int row_first(int t) { return row(t, 0); }
    static constexpr int row_last(int t) { return row(t, cells(t) - 1); }
    static constexpr int col_first(int t) { return col(t, 0); }
    static constexpr int col_last(int t) { return ROWS ? (t == 0 ? 0 : 8) : (t == 0 ? OW - 1 : 2 * t - 1); }
    // kernel row kh / kernel column kw of tile t reads at least one pixel
    static constexpr bool live_kh(int t, int kh) { return row_last(t) + kh - 2 >= 0 && row_first(t) + kh - 2 < IH; }
    static constexpr bool live_kw(int t, int kw) { return col_last(t) + kw - 2 >= 0 && col_first(t) + kw - 2 < IW; }
    static constexpr int kw_first(int t) { return live_kw(t, 0) ? 0 : live_kw(t, 1) ? 1 : 2; }    // the live ones are contiguous
    static constexpr int kw_last(int t) { return live_kw(t, 3) ? 3 : live_kw(t, 2) ? 2 : 1; }
    static constexpr bool every_kw(int t) { return kw_first(t) == 0 && kw_last(t) == 3; }
    static constexpr int pairs() {
        int n = 0;
        for (int t = 0; t < NT; ++t)
            for (int k = 0; k < 16; ++k) n += live_kh(t, k >> 2) && live_kw(t, k & 3);
        return n;
    }
    // every output cell sits in exactly one tile, and no tap a cell needs is skipped for its tile
    static constexpr bool sound() {
        int seen[OH * OW] = {};
        for (int t = 0; t < NT; ++t)
            for (int i = 0; i < cells(t); ++i) {
                const int y = row(t, i), x = col(t, i);
                if (y < 0 || y >= OH || x < 0 || x >= OW) return false;
                ++seen[y * OW + x];
                for (int k = 0; k < 16; ++k) {
                    const bool reads_pixel = y + (k >> 2) - 2 >= 0 && y + (k >> 2) - 2 < IH && x + (k & 3) - 2 >= 0 && x + (k & 3) - 2 < IW;
                    if (reads_pixel && !(live_kh(t, k >> 2) && live_kw(t, k & 3))) return false;
                }
            }
        for (int c = 0; c < OH * OW; ++c)
            if (seen[c] != 1) return false;
        return true;
    }
};
static_assert(BoardTiles<7, 8>::NT == 5 && BoardTiles<7, 8>::pairs() == 64 && BoardTiles<7, 8>::sound(), "7x8 -> 8x9");
static_assert(BoardTiles<8, 9>::NT == 6 && BoardTiles<8, 9>::pairs() == 80 && BoardTiles<8, 9>::sound(), "8x9 -> 9x10");

// All four taps of kernel row KH for the live tiles (a tile whose cells all sit in a border column skips the kernel
// columns that read nothing but padding: one uniform branch per k-step).  Fragments of one (tap, k-step) are read one step ahead into the
// other of two register sets; weights BRING - 1 steps ahead.  acc[t][j]: tile t, channel tile j.
template <int IH, int IW, int NT, int BRING, int KH>
__device__ __forceinline__ void conv_board_row(f32x4 (&acc)[NT][2], u32x4 (&bh)[BRING][2], u32x4 (&bl)[BRING][2],
                                               const unsigned char *smem, const int (&tbase)[NT],
                                               const __amdgpu_buffer_rsrc_t &rwh, const __amdgpu_buffer_rsrc_t &rwl, int woff) {
    using G = BoardTiles<IH, IW>;
    static_assert(NT == G::NT, "one accumulator pair per tile");
    constexpr int W2 = G::W2, PB = SP_PB;
    u32x4 ah[2][NT], al[2][NT];
    const unsigned char *p0 = smem + ((KH - 2) * W2 - 2) * PB;
    static_for<NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        if constexpr (G::live_kh(t, KH)) {
            ah[0][t] = *reinterpret_cast<const u32x4 *>(p0 + tbase[t]);
            al[0][t] = *reinterpret_cast<const u32x4 *>(p0 + tbase[t] + SP_LO);
        }
    });
#pragma unroll 1
    for (int kw = 0; kw < 4; ++kw) {
        const unsigned char *pv = p0 + kw * PB;
        const bool more = kw < 3;
        const int step0 = (KH * 4 + kw) * 4;
        static_for<4>([&](auto Kc) __attribute__((always_inline)) {
            constexpr int ks = decltype(Kc)::value;
            constexpr int cur = ks & 1, nxt = cur ^ 1;
            constexpr int slot = ks % BRING, fill = (ks + BRING - 1) % BRING;
            static_assert(4 % BRING == 0, "the weight ring must divide the k-steps of a tap");
#ifndef NNB_ABL_NO_CONV_WEIGHT_LOADS                              // (timing experiments only: tools/split_bench.hip)
            {
                const int sn = (step0 + ks + BRING - 1) * 8192;      // past the layer's last step: zeros, never used
                bh[fill][0] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, sn, 0);
                bh[fill][1] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff + 256, sn, 0);
                bl[fill][0] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, sn, 0);
                bl[fill][1] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff + 256, sn, 0);
            }
#endif
            static_for<NT>([&](auto T) __attribute__((always_inline)) {       // fragments of the next step
                constexpr int t = decltype(T)::value;
                if constexpr (G::live_kh(t, KH)) {
                    if constexpr (ks < 3) {
                        ah[nxt][t] = *reinterpret_cast<const u32x4 *>(pv + tbase[t] + (ks + 1) * 64);
                        al[nxt][t] = *reinterpret_cast<const u32x4 *>(pv + tbase[t] + (ks + 1) * 64 + SP_LO);
                    } else if (more) {
                        ah[nxt][t] = *reinterpret_cast<const u32x4 *>(pv + PB + tbase[t]);
                        al[nxt][t] = *reinterpret_cast<const u32x4 *>(pv + PB + tbase[t] + SP_LO);
                    }
                }
            });
            const half8 wh0 = as_half8(bh[slot][0]), wh1 = as_half8(bh[slot][1]);
            const half8 wl0 = as_half8(bl[slot][0]), wl1 = as_half8(bl[slot][1]);
            static_for<NT>([&](auto T) __attribute__((always_inline)) {
                constexpr int t = decltype(T)::value;
                if (G::live_kh(t, KH) && (G::every_kw(t) || (kw >= G::kw_first(t) && kw <= G::kw_last(t)))) {
                    const half8 xh = as_half8(ah[cur][t]), xl = as_half8(al[cur][t]);
                    // per accumulator: hi*lo, lo*hi, hi*hi, in this order; the two channel tiles alternate
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl1, acc[t][1], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh1, acc[t][1], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh1, acc[t][1], 0, 0, 0);
                }
            });
        });
    }
}

// One 128->128 layer over the board's image at `smem`: accumulators cleared, weights primed, four kernel rows.
template <int IH, int IW, int NT, int BRING>
__device__ __forceinline__ void conv_board_layer(f32x4 (&acc)[NT][2], const unsigned char *smem,
                                                 const u32x4 *wph, const u32x4 *wpl, int lane, int wave) {
    using G = BoardTiles<IH, IW>;
    static_assert(NT == G::NT, "one accumulator pair per tile");
    const int m = lane & 15, kg = lane >> 4;
    int tbase[G::NT];
    static_for<G::NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        const int i = m < G::cells(t) ? m : G::cells(t) - 1;     // rows past the tile's last cell repeat it
        tbase[t] = (G::LEAD + G::row(t, i) * G::W2 + G::col(t, i)) * SP_PB + kg * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; }
    });
    const __amdgpu_buffer_rsrc_t rwh = conv_weight_rsrc(wph), rwl = conv_weight_rsrc(wpl);
    const int woff = (kg * 128 + wave * 32 + m) * 16;           // + 256 bytes: the second channel tile
    u32x4 bh[BRING][2], bl[BRING][2];
#pragma unroll
    for (int i = 0; i < BRING - 1; ++i) {
        bh[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, i * 8192, 0);
        bh[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff + 256, i * 8192, 0);
        bl[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, i * 8192, 0);
        bl[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff + 256, i * 8192, 0);
    }
    conv_board_row<IH, IW, NT, BRING, 0>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    conv_board_row<IH, IW, NT, BRING, 1>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    conv_board_row<IH, IW, NT, BRING, 2>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    conv_board_row<IH, IW, NT, BRING, 3>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
}

// v = ReLU(acc * k + bias) in place; returns this lane's maximum over the board's cells
template <typename G>
__device__ __forceinline__ float conv_board_finish(f32x4 (&acc)[G::NT][2], float k, const float (&bv)[2], int kg) {
    float mx = 0.f;
    static_for<G::NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool valid = G::cells(t) == 16 || 4 * kg + r < G::cells(t);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v = fmaxf(__builtin_fmaf(acc[t][j][r], k, bv[j]), 0.f);
                acc[t][j][r] = v;
                mx = fmaxf(mx, valid ? v : 0.f);
            }
        }
    });
    return mx;
}

// Quantised with the board's scale, hi / lo into LDS: the output cell at (row, column) goes to position
// LEADO + row * RSO + column.  A lane swaps its (hi, lo) pair with the neighbouring channel's lane (same cell): the
// even lane stores both hi halves, the odd lane both lo halves, one 32-bit word each (as conv_split_store).
template <typename G, int RSO, int LEADO>
__device__ __forceinline__ void conv_board_store(const f32x4 (&acc)[G::NT][2], float sc_out, unsigned *ldsw, int wave, int lane) {
    const int m = lane & 15, kg = lane >> 4, odd = lane & 1;
    const unsigned psel = odd ? 0x03020706u : 0x05040100u;
    static_for<G::NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * kg + r;
            const int pos = LEADO + G::row(t, i) * RSO + G::col(t, i);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                half2v pr;
                const float x = acc[t][j][r] * sc_out;
                pr[0] = (_Float16)x;
                pr[1] = (_Float16)(x - (float)pr[0]);
                const unsigned own = __builtin_bit_cast(unsigned, pr);
                const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // lane ^ 1
                const int ch = wave * 32 + 16 * j + m;
                if (G::cells(t) == 16 || i < G::cells(t))
                    ldsw[pos * (SP_PB / 4) + (ch >> 1) + (odd ? SP_LO / 4 : 0)] = __builtin_amdgcn_perm(other, own, psel);
            }
        }
    });
}

// Leaf encode (AlphaZeroAgent.preprocess, agent_code.py:76-89) + the 2->128 layer (network.py:29-30) into the image of
// the 7x8 -> 8x9 layer at `smem_raw`, by a block of 256 threads (wave w forms output channels 32w .. 32w+31); the caller
// synchronises the block afterwards.
__device__ __forceinline__ void conv_first_image(unsigned char *smem_raw, const u64 *__restrict__ side, int board,
                                                 const u32x4 *__restrict__ w0h, const u32x4 *__restrict__ w0l,
                                                 const float *__restrict__ b0, const SplitScales *__restrict__ sc, int tid) {
    using G1 = BoardGeom<7, 8>;
    constexpr int PB = SP_PB;
    constexpr int IW1 = 8, IP1 = 56, W21 = G1::W2, LEAD1 = G1::LEAD, NPOS1 = G1::NPOS;
    u32x4 *lds = reinterpret_cast<u32x4 *>(smem_raw);
    const int lane = tid & 63, nt = tid >> 6, m = lane & 31, h = lane >> 5;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // ---- leaf encode (AlphaZeroAgent.preprocess, agent_code.py:76-89) + 2->128 layer (network.py:29-30) into the first
    //      image, with the layer's static output scale: the two planes are expanded from the bitboards into the A operand
    //      of 32x32x16 tiles of 32 cells (k = plane*16 + kh*4 + kw; lane half h holds k%16 in [8h, 8h+8)) ----
    for (int e = tid; e < NPOS1 * SP_RS; e += 256) lds[e] = zero4;
    __syncthreads();
    {
        half8 bwh[2], bwl[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            bwh[pl] = as_half8(w0h[(nt * 32 + m) * 4 + 2 * pl + h]);
            bwl[pl] = as_half8(w0l[(nt * 32 + m) * 4 + 2 * pl + h]);
        }
        const float bv_in = b0[nt * 32 + m];
        const float inv_sw0 = sc->inv_sw0, sa1 = sc->sa1;
        _Float16 *ldsh_in = reinterpret_cast<_Float16 *>(smem_raw);
        const u64 planes[2] = {side[2 * board], side[2 * board + 1]};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int pos = mt * 32 + m, oh = pos >> 3, ow = pos & 7;
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                half8 a;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int ih = oh + 2 * h + rr - 2;
                    u64 t = (unsigned)ih < 6u ? (planes[pl] >> (5 - ih)) & c4::kBottomMask : 0;
                    unsigned row = (unsigned)((t * 0x41041041040ULL) >> 42) & 0x7F;
                    unsigned nib = ((row << 2) >> ow) & 0xF;
#pragma unroll
                    for (int kw = 0; kw < 4; ++kw) a[rr * 4 + kw] = ((nib >> kw) & 1) ? (_Float16)1.0f : (_Float16)0.0f;
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bwl[pl], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bwh[pl], c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p < IP1) {
                    const float v = fmaxf(__builtin_fmaf(c[r], inv_sw0, bv_in), 0.f);
                    _Float16 hi, lo;
                    split_half(v * sa1, hi, lo);
                    _Float16 *dst = ldsh_in + (LEAD1 + (p / IW1) * W21 + p % IW1) * (PB / 2) + nt * 32 + m;
                    dst[0] = hi;
                    dst[128] = lo;
                }
            }
        }
    }
}

template <int BRING, bool WRITE_MID>
__global__ __launch_bounds__(256, 2) void conv_board_split_kernel(const u64 *__restrict__ side,
                                                                  const u32x4 *__restrict__ w0h, const u32x4 *__restrict__ w0l,
                                                                  const float *__restrict__ b0,
                                                                  const u32x4 *__restrict__ w2h, const u32x4 *__restrict__ w2l,
                                                                  const float *__restrict__ bias2,
                                                                  const u32x4 *__restrict__ w4h, const u32x4 *__restrict__ w4l,
                                                                  const float *__restrict__ bias4,
                                                                  const SplitScales *__restrict__ sc,
                                                                  u32x4 *__restrict__ mid, float *__restrict__ mid_inv_scale,
                                                                  u32x4 *__restrict__ out, float *__restrict__ out_inv_scale,
                                                                  const int32_t *__restrict__ count_ptr, int n_fixed,
                                                                  unsigned long long *__restrict__ clk) {
    using G1 = BoardTiles<7, 8>;
    using G2 = BoardTiles<8, 9>;
    constexpr int PB = SP_PB;
    constexpr int NPOS1 = BoardGeom<7, 8>::NPOS;
    constexpr int IH2 = 8, IW2 = 9, IP2 = 72, W22 = G2::W2, LEAD2 = G2::LEAD, NPOS2 = BoardGeom<8, 9>::NPOS;
    constexpr int OP2 = G2::NC;
    constexpr int REGION = (NPOS2 > NPOS1 ? (NPOS2 > OP2 ? NPOS2 : OP2) : (NPOS1 > OP2 ? NPOS1 : OP2)) * PB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4 *lds = reinterpret_cast<u32x4 *>(smem_raw);
    float *s_max = reinterpret_cast<float *>(smem_raw + REGION);     // [4 waves]

    const int count = batch_count_sp(count_ptr, n_fixed);
    const int board = blockIdx.x;
    if (board >= count) return;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    const int tid = threadIdx.x;
    const int lane = tid & 63, nt = tid >> 6;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    NNB_SP_STAMP(0) NNB_SP_WALL(5)

    conv_first_image(smem_raw, side, board, w0h, w0l, b0, sc, tid);
    __syncthreads();
    NNB_SP_STAMP(1)

    const int m16 = lane & 15, kg = lane >> 4;
    float k2;                                                    // 1 / (scale of the middle activations * weight scale)
    // ---- first 128->128 layer ----
    {
        f32x4 acc[G1::NT][2];
        conv_board_layer<7, 8, G1::NT, BRING>(acc, smem_raw, w2h, w2l, lane, nt);
        NNB_SP_STAMP(2)
        const float k1 = sc->inv_sa1 * sc->inv_sw2;
        const float bv[2] = {bias2[nt * 32 + m16], bias2[nt * 32 + 16 + m16]};
        float v = conv_board_finish<G1>(acc, k1, bv, kg);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
        if (lane == 0) s_max[nt] = v;
        __syncthreads();                                         // every wave is done with the first image
        float sinv;
        const float sout = pow2_scale(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])), sinv);
        // the second image: zeros wherever there is no pixel, the quantised activations where there is one
        constexpr int TOP = LEAD2, BOT = NPOS2 - (LEAD2 + IH2 * W22), NPAD = TOP + 2 * IH2 + BOT;
        static_assert(NPAD + IP2 == NPOS2, "every position of the image is a pixel or padding");
        for (int e = tid; e < NPAD * 32; e += 256) {
            const int kk = e >> 5, q = e & 31;
            const int pos = kk < TOP ? kk
                          : kk < TOP + 2 * IH2 ? LEAD2 + ((kk - TOP) >> 1) * W22 + IW2 + ((kk - TOP) & 1)
                                               : LEAD2 + IH2 * W22 + (kk - TOP - 2 * IH2);
            lds[pos * SP_RS + q] = zero4;
        }
        conv_board_store<G1, W22, LEAD2>(acc, sout, reinterpret_cast<unsigned *>(smem_raw), nt, lane);
        k2 = sinv * sc->inv_sw4;
        if constexpr (WRITE_MID) {
            if (tid == 0) mid_inv_scale[board] = sinv;
        }
        __syncthreads();                                         // the second image is complete (s_max may be reused)
        if constexpr (WRITE_MID) {
            u32x4 *dst = mid + (size_t)board * IP2 * 32;
            for (int e = tid; e < IP2 * 32; e += 256) {
                const int p = e >> 5, q = e & 31;
                dst[e] = lds[(LEAD2 + (p / IW2) * W22 + p % IW2) * SP_RS + q];
            }
        }
    }
    NNB_SP_STAMP(4)

    // ---- second 128->128 layer ----
    f32x4 acc[G2::NT][2];
    conv_board_layer<8, 9, G2::NT, BRING>(acc, smem_raw, w4h, w4l, lane, nt);
    NNB_SP_STAMP(7)
    const float bv[2] = {bias4[nt * 32 + m16], bias4[nt * 32 + 16 + m16]};
    float v = conv_board_finish<G2>(acc, k2, bv, kg);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    if (lane == 0) s_max[nt] = v;
    __syncthreads();                                             // every wave is done with the second image
    float sinv;
    const float sout = pow2_scale(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])), sinv);
    conv_board_store<G2, G2::OW, 0>(acc, sout, reinterpret_cast<unsigned *>(smem_raw), nt, lane);
    __syncthreads();
    u32x4 *dst = out + (size_t)board * OP2 * 32;
    for (int e = tid; e < OP2 * 32; e += 256) dst[e] = lds[(e >> 5) * SP_RS + (e & 31)];
    if (tid == 0) out_inv_scale[board] = sinv;
    NNB_SP_STAMP(3) NNB_SP_WALL(6)
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

// ---------------------------------------------------------------------------
// The same convolutions for a HANDFUL of boards (at most kSpreadMaxBoards = 128: single searches, small tournaments), where
// only latency counts: conv_board_split_kernel gives a board to one block, whose four waves each walk 5 or 6 tiles x 2
// channel tiles through 64 weight steps - about 50 us however few boards there are.  Here FOUR blocks share a board and
// every wave owns one channel tile of 16 and half of the board's cell tiles, in three launches:
//   conv_spread_first_kernel   leaf encode + 2->128 (every block, whole image) and its 32 channels of the 7x8 -> 8x9
//                              layer: ReLU(acc * k + bias) as float32 to `mid_f32`, the board's maximum by atomic max
//   conv_spread_second_kernel  the 8x9 image from `mid_f32` quantised with the board's scale, its 32 channels of the
//                              8x9 -> 9x10 layer to `out_f32`, the maximum likewise
//   conv_spread_pack_kernel    `out_f32` quantised with the board's scale into the linear layer's input; clears the maxima
// Every accumulator sums the products conv_board_split_kernel sums, in its order (kernel rows, taps, k-steps ascending;
// hi*lo, lo*hi, hi*hi), a maximum does not depend on the order it is taken in, and the scales and the hi / lo split are
// the same functions of the same float32 values: the outputs equal the one-block form's bit for bit.
// ---------------------------------------------------------------------------
constexpr int kSpreadMaxBoards = kConvSpreadMaxBoards;
constexpr int kSpreadMidFloats = 72 * 128, kSpreadOutFloats = 90 * 128;
constexpr int kSpreadFloatsPerBoard = kSpreadMidFloats + kSpreadOutFloats;
static_assert(kSpreadFloatsPerBoard == kConvSpreadFloatsPerBoard, "the workspace holds this many floats per board");

// Kernel row KH for tiles T0 .. T0+TN-1 and ONE channel tile (cf. conv_board_row).  A wave has few matrix instructions
// per weight step here (at most 9), so the weights run a whole kernel row (16 steps) ahead to cover the L2 latency,
// and the row is unrolled completely.
template <int IH, int IW, int T0, int TN, int KH>
__device__ __forceinline__ void conv_spread_row(f32x4 (&acc)[TN], u32x4 (&bh)[16], u32x4 (&bl)[16],
                                                const unsigned char *smem, const int (&tbase)[TN],
                                                const __amdgpu_buffer_rsrc_t &rwh, const __amdgpu_buffer_rsrc_t &rwl, int woff) {
    using G = BoardGeom<IH, IW>;
    constexpr int W2 = G::W2, PB = SP_PB;
    u32x4 ah[2][TN], al[2][TN];
    const unsigned char *p0 = smem + ((KH - 2) * W2 - 2) * PB;
    static_for<TN>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        if constexpr (G::live(T0 + t, KH)) {
            ah[0][t] = *reinterpret_cast<const u32x4 *>(p0 + tbase[t]);
            al[0][t] = *reinterpret_cast<const u32x4 *>(p0 + tbase[t] + SP_LO);
        }
    });
    static_for<16>([&](auto Sc) __attribute__((always_inline)) {
        constexpr int st = decltype(Sc)::value, kw = st >> 2, ks = st & 3;
        constexpr int cur = st & 1, nxt = cur ^ 1;
        const u32x4 wh4 = bh[st], wl4 = bl[st];
        if constexpr (KH < 3) {                                       // the same step of the next kernel row
            bh[st] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, ((KH + 1) * 16 + st) * 8192, 0);
            bl[st] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, ((KH + 1) * 16 + st) * 8192, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                            // the loads stay a row ahead of their use
        static_for<TN>([&](auto T) __attribute__((always_inline)) {       // fragments of the next step
            constexpr int t = decltype(T)::value;
            if constexpr (G::live(T0 + t, KH) && st < 15) {
                constexpr int off = ks < 3 ? kw * PB + (ks + 1) * 64 : (kw + 1) * PB;
                ah[nxt][t] = *reinterpret_cast<const u32x4 *>(p0 + off + tbase[t]);
                al[nxt][t] = *reinterpret_cast<const u32x4 *>(p0 + off + tbase[t] + SP_LO);
            }
        });
        const half8 wh = as_half8(wh4), wl = as_half8(wl4);
        static_for<TN>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            if constexpr (G::live(T0 + t, KH)) {
                const half8 xh = as_half8(ah[cur][t]), xl = as_half8(al[cur][t]);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh, acc[t], 0, 0, 0);
            }
        });
    });
}

// One wave's share of a 128->128 layer: tiles T0 .. T0+TN-1, channel tile `ctile` (16 channels), then v = ReLU(acc * k +
// bias) as float32 to dst[cell][128]; returns the lane's maximum over the valid cells.
template <int IH, int IW, int T0, int TN>
__device__ __forceinline__ float conv_spread_layer(const unsigned char *smem, const u32x4 *wph, const u32x4 *wpl,
                                                   const float *__restrict__ bias, float k, float *__restrict__ dst,
                                                   int lane, int ctile) {
    using G = BoardGeom<IH, IW>;
    static_assert(T0 + TN <= G::NT, "tiles of this layer");
    const int m = lane & 15, kg = lane >> 4;
    int tbase[TN];
    f32x4 acc[TN];
    static_for<TN>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        int c = 16 * (T0 + t) + m;
        c = c < G::NC ? c : G::NC - 1;                           // rows past the last cell repeat it
        tbase[t] = (G::LEAD + (c / G::OW) * G::W2 + c % G::OW) * SP_PB + kg * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    });
    const __amdgpu_buffer_rsrc_t rwh = conv_weight_rsrc(wph), rwl = conv_weight_rsrc(wpl);
    const int woff = (kg * 128 + ctile * 16 + m) * 16;
    u32x4 bh[16], bl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {                              // the first kernel row's weights
        bh[i] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, i * 8192, 0);
        bl[i] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, i * 8192, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    conv_spread_row<IH, IW, T0, TN, 0>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    conv_spread_row<IH, IW, T0, TN, 1>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    conv_spread_row<IH, IW, T0, TN, 2>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    conv_spread_row<IH, IW, T0, TN, 3>(acc, bh, bl, smem, tbase, rwh, rwl, woff);
    const float bv = bias[ctile * 16 + m];
    float mx = 0.f;
    static_for<TN>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (T0 + t) + 4 * kg + r;
            const float v = fmaxf(__builtin_fmaf(acc[t][r], k, bv), 0.f);
            if (c < G::NC) {
                dst[c * 128 + ctile * 16 + m] = v;
                mx = fmaxf(mx, v);
            }
        }
    });
    return mx;
}

// The block's maximum (non-negative floats compare as their bit patterns) into the board's word.
__device__ __forceinline__ void conv_spread_block_max(float v, float *s_max, unsigned *__restrict__ word, int lane, int wave) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    if (lane == 0) s_max[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(word, __float_as_uint(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]))));
}

constexpr size_t conv_spread_lds_bytes() {
    int a = conv_board_positions(7, 8), b = conv_board_positions(8, 9);
    return (size_t)(a > b ? a : b) * SP_PB + 64;
}

__global__ __launch_bounds__(256, 2) void conv_spread_first_kernel(const u64 *__restrict__ side,
                                                                   const u32x4 *__restrict__ w0h, const u32x4 *__restrict__ w0l,
                                                                   const float *__restrict__ b0,
                                                                   const u32x4 *__restrict__ w2h, const u32x4 *__restrict__ w2l,
                                                                   const float *__restrict__ bias2,
                                                                   const SplitScales *__restrict__ sc,
                                                                   float *__restrict__ f32_scratch, unsigned *__restrict__ maxima,
                                                                   const int32_t *__restrict__ count_ptr, int n_fixed,
                                                                   unsigned long long *__restrict__ clk) {
    using G1 = BoardGeom<7, 8>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *s_max = reinterpret_cast<float *>(smem_raw + conv_spread_lds_bytes() - 64);
    const int count = batch_count_sp(count_ptr, n_fixed);
    const int board = blockIdx.x >> 2, cg = blockIdx.x & 3;
    if (board >= count) return;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    conv_first_image(smem_raw, side, board, w0h, w0l, b0, sc, tid);
    __syncthreads();
    const float k1 = sc->inv_sa1 * sc->inv_sw2;
    float *dst = f32_scratch + (size_t)board * kSpreadFloatsPerBoard;
    const int ctile = cg * 2 + (wave >> 1);
    static_assert(G1::NT == 5, "tiles 0-2 and 3-4");
    const float mx = (wave & 1) == 0 ? conv_spread_layer<7, 8, 0, 3>(smem_raw, w2h, w2l, bias2, k1, dst, lane, ctile)
                                     : conv_spread_layer<7, 8, 3, 2>(smem_raw, w2h, w2l, bias2, k1, dst, lane, ctile);
    conv_spread_block_max(mx, s_max, maxima + 2 * board, lane, wave);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

template <bool WRITE_MID>
__global__ __launch_bounds__(256, 2) void conv_spread_second_kernel(const u32x4 *__restrict__ w4h, const u32x4 *__restrict__ w4l,
                                                                    const float *__restrict__ bias4,
                                                                    const SplitScales *__restrict__ sc,
                                                                    float *__restrict__ f32_scratch, unsigned *__restrict__ maxima,
                                                                    u32x4 *__restrict__ mid, float *__restrict__ mid_inv_scale,
                                                                    const int32_t *__restrict__ count_ptr, int n_fixed) {
    using G2 = BoardGeom<8, 9>;
    constexpr int IH2 = 8, IW2 = 9, IP2 = 72, W22 = G2::W2, LEAD2 = G2::LEAD, NPOS2 = G2::NPOS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4 *lds = reinterpret_cast<u32x4 *>(smem_raw);
    float *s_max = reinterpret_cast<float *>(smem_raw + conv_spread_lds_bytes() - 64);
    const int count = batch_count_sp(count_ptr, n_fixed);
    const int board = blockIdx.x >> 2, cg = blockIdx.x & 3;
    if (board >= count) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    float *scratch = f32_scratch + (size_t)board * kSpreadFloatsPerBoard;
    float sinv;
    const float sout = pow2_scale(__uint_as_float(maxima[2 * board]), sinv);
    // the 8x9 image: zeros wherever there is no pixel, the quantised activations where there is one
    constexpr int TOP = LEAD2, BOT = NPOS2 - (LEAD2 + IH2 * W22), NPAD = TOP + 2 * IH2 + BOT;
    static_assert(NPAD + IP2 == NPOS2, "every position of the image is a pixel or padding");
    for (int e = tid; e < NPAD * 32; e += 256) {
        const int kk = e >> 5, q = e & 31;
        const int pos = kk < TOP ? kk
                      : kk < TOP + 2 * IH2 ? LEAD2 + ((kk - TOP) >> 1) * W22 + IW2 + ((kk - TOP) & 1)
                                           : LEAD2 + IH2 * W22 + (kk - TOP - 2 * IH2);
        lds[pos * SP_RS + q] = zero4;
    }
    {
        unsigned *ldsw = reinterpret_cast<unsigned *>(smem_raw);
        const float2 *src = reinterpret_cast<const float2 *>(scratch);
        for (int e = tid; e < IP2 * 64; e += 256) {              // one channel pair of one pixel
            const int p = e >> 6, q = e & 63;
            const float2 v = src[e];
            _Float16 h0, l0, h1, l1;
            split_half(v.x * sout, h0, l0);
            split_half(v.y * sout, h1, l1);
            const int pos = LEAD2 + (p / IW2) * W22 + p % IW2;
            ldsw[pos * (SP_PB / 4) + q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            ldsw[pos * (SP_PB / 4) + SP_LO / 4 + q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
    }
    __syncthreads();
    if constexpr (WRITE_MID) {
        if (cg == 0) {
            if (tid == 0) mid_inv_scale[board] = sinv;
            u32x4 *dstm = mid + (size_t)board * IP2 * 32;
            for (int e = tid; e < IP2 * 32; e += 256) {
                const int p = e >> 5, q = e & 31;
                dstm[e] = lds[(LEAD2 + (p / IW2) * W22 + p % IW2) * SP_RS + q];
            }
        }
    }
    const float k2 = sinv * sc->inv_sw4;
    float *dst = scratch + kSpreadMidFloats;
    const int ctile = cg * 2 + (wave >> 1);
    static_assert(G2::NT == 6, "tiles 0-2 and 3-5");
    const float mx = (wave & 1) == 0 ? conv_spread_layer<8, 9, 0, 3>(smem_raw, w4h, w4l, bias4, k2, dst, lane, ctile)
                                     : conv_spread_layer<8, 9, 3, 3>(smem_raw, w4h, w4l, bias4, k2, dst, lane, ctile);
    conv_spread_block_max(mx, s_max, maxima + 2 * board + 1, lane, wave);
}

__global__ __launch_bounds__(256) void conv_spread_pack_kernel(const float *__restrict__ f32_scratch, unsigned *__restrict__ maxima,
                                                               u32x4 *__restrict__ out, float *__restrict__ out_inv_scale,
                                                               const int32_t *__restrict__ count_ptr, int n_fixed) {
    const int count = batch_count_sp(count_ptr, n_fixed);
    const int board = blockIdx.x, tid = threadIdx.x;
    if (board >= count) return;
    float sinv;
    const float sout = pow2_scale(__uint_as_float(maxima[2 * board + 1]), sinv);
    const float2 *src = reinterpret_cast<const float2 *>(f32_scratch + (size_t)board * kSpreadFloatsPerBoard + kSpreadMidFloats);
    unsigned *dst = reinterpret_cast<unsigned *>(out + (size_t)board * 90 * 32);
    for (int e = tid; e < 90 * 64; e += 256) {
        const int p = e >> 6, q = e & 63;
        const float2 v = src[e];
        _Float16 h0, l0, h1, l1;
        split_half(v.x * sout, h0, l0);
        split_half(v.y * sout, h1, l1);
        dst[p * 128 + q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        dst[p * 128 + 64 + q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
    __syncthreads();                                             // every thread has its scale: the maxima start the next pass at zero
    if (tid == 0) { out_inv_scale[board] = sinv; maxima[2 * board] = 0u; maxima[2 * board + 1] = 0u; }
}

// ---------------------------------------------------------------------------
// linear 11520->1024 (network.py:36-39,46-47) with split operands, split in two halves of K.
// Block tile (32*MTR) rows x 256 columns, 4 waves (one per SIMD); wave w owns columns 64w..64w+63
// (two 32-column tiles) for all row tiles, so one k-step of 16 is 8 fragment reads from LDS (hi and
// lo of up to 4 row tiles), 4 weight loads from L2 (hi and lo of 2 column tiles) and up to 24 MFMAs.
// A slabs of 64 k (128 bytes of hi and 128 of lo per row) go global -> registers four slabs ahead and
// registers -> LDS two slabs ahead (three LDS buffers of 32 KB, one barrier per slab).  The two K
// halves of a tile meet in fc_split_finish, which adds them, the bias and the ReLU and reduces the
// row against the head rows.  Blocks that share blockIdx % 8 (one XCD, as blocks are dealt out
// today: a matter of speed only) cover 8 row blocks x (2 column blocks x 2 K halves), so the two
// halves of a tile share an L2 and an L2 fetches half the weight matrix.
// ---------------------------------------------------------------------------
constexpr int FCS_K = 11520, FCS_N = 1024, FCS_BK = 64, FCS_SPLIT = 2;
constexpr int FCS_ROW_BYTES = 90 * 512;                      // one activation row: 90 cells x (128 hi + 128 lo halves)
// LDS slab buffer: 128 rows of 16 pieces of 16 bytes (8 hi, 8 lo) + one pad piece, so that the 16 rows one
// ds_read_b128 services together fall on 16 different bank groups and every fragment address is the lane's
// base plus a compile-time offset.
[[maybe_unused]] constexpr int FCS_BR = 6;                   // weight fragments in flight, in k-steps of 16 (+1): fc_split_kernel
[[maybe_unused]] constexpr int FC16_BR = 3;                  // the same in k-steps of 32: fc16_kernel
constexpr int FCS_LROW = 17 * 16;                            // bytes per row in LDS
constexpr int FCS_LBUF = 128 * FCS_LROW;                     // bytes per buffer

// ---------------------------------------------------------------------------
// Last step of the linear layer (network.py:46-49 up to the heads' sums), shared by the large- and small-batch
// forms.  The two blocks that hold the two K halves of one output tile meet here: each draws a ticket; the one
// that arrives FIRST hands its partial sums (true units) to the other through memory and leaves, the SECOND adds
// them to its own, adds the bias, applies the ReLU and multiplies the row by the head rows, so that per leaf
// 8 x 32 (or, under the reference rule, which ignores priors, 32) floats leave the kernel instead of 2 x 1024.
// The second never waits for a block that has not started: a first arriver is running by construction.
//
// Hand-off (blocks may sit on different XCDs, whose L2s are not coherent): the first writes the tile with
// write-through 16-byte stores in its own register layout (the partner has the same), every storing wave drains
// its stores, one lane raises the pair's flag to the ticket it drew + 1; the second polls that word relaxed,
// then ONE agent-scope acquire drops its CU's stale lines before plain loads.  Tickets only grow (two per launch
// and pair), so nothing has to be cleared between launches: even ticket = first, and the flag the second waits
// for is its own ticket.
//
// A head's dot product over a tile's 32 columns is a balanced tree over the column index (column ^ 16 first, then
// ^ 8, ^ 4, ^ 2, ^ 1), whatever the batch size or kernel form, and heads_z_from_q adds the 32 tiles the same way.
// ---------------------------------------------------------------------------
struct FcFinish {
    const float *fc_bias;       // [1024]
    const float *heads_w;       // [8][1024]: policy rows 0..6, value row 7
    float *q;                   // [rows][8][32]
    float *hidden;              // [rows][1024] or null
    float *handoff;             // [(max_rows + 128) * 1024]
    unsigned *ticket, *flag;    // [pairs] each
    int32_t *err;               // += 1 when a wait for the partner's tile runs out (results are then invalid)
    int heads_all;              // 0: the value head only
};

// Value of lane ^ K for K = 1, 2, 4, 8, in the vector pipe (DPP): quad permutes, a rotation by 8 within a row of 16
// lanes, and for K = 4 a shift left into lane groups 0 and 2 of a row plus a shift right into groups 1 and 3.
template <int K>
__device__ __forceinline__ float lane_xor(float v) {
    const int x = __float_as_int(v);
    int r;
    if constexpr (K == 1) r = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);             // quad_perm [1,0,3,2]
    else if constexpr (K == 2) r = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);        // quad_perm [2,3,0,1]
    else if constexpr (K == 4) {
        r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false);                       // row_shl:4 -> groups 0, 2
        r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);                       // row_shr:4 -> groups 1, 3
    } else {
        static_assert(K == 8, "lane distances 1, 2, 4 and 8");
        r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);                       // row_ror:8
    }
    return __int_as_float(r);
}

// p[r], r = 0..15: this lane's column of an MFMA tile's 16 rows.  Returns the sum over the 32 columns (lanes of
// this half wave) of ONE of those rows: r = 8*bit4(m) + 4*bit3(m) + 2*bit2(m) + bit1(m).  Each exchange halves the
// values a lane carries, so 16 exchanges do what 80 would do by reducing every row in every lane.  The first level
// (lanes 16 apart) is v_permlane16_swap: it trades the odd 16-lane rows of p[i] for the even rows of p[8 + i], after
// which every lane holds its own value and its partner's for the row it keeps (a + b = b + a bit for bit, so which of
// the two arrived does not matter).  tools/lane_exchange_check.hip checks all five exchanges on the device.
__device__ __forceinline__ float tile_column_sum(const float (&p)[16], int m) {
    float v8[8], v4[4], v2[2];
    const bool b3 = m & 8, b2 = m & 4, b1 = m & 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[i]), __float_as_uint(p[8 + i]), false, false);
        v8[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = (b3 ? v8[4 + i] : v8[i]) + lane_xor<8>(b3 ? v8[i] : v8[4 + i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) v2[i] = (b2 ? v4[2 + i] : v4[i]) + lane_xor<4>(b2 ? v4[i] : v4[2 + i]);
    const float v1 = (b1 ? v2[1] : v2[0]) + lane_xor<2>(b1 ? v2[0] : v2[1]);
    return v1 + lane_xor<1>(v1);
}

// acc[i], i = 0..NT-1: tile i covers rows row0 + 32*(i / NU) .. + 31 and column tile tile0 + i % NU (32 columns);
// on entry the raw accumulators.  `wave` = this wave's index in the block, `ticket` = what the block drew for `pair`
// (uniform).  hand_base: float offset of the pair's hand-off area (NW * NT * 1024 floats, NW = waves per block).
template <int NT, int NU, int NW>
__device__ __forceinline__ void fc_split_finish(f32x16 (&acc)[NT], const FcFinish &fin, const float *__restrict__ a_inv_scale,
                                                float inv_sw, unsigned ticket, int pair, size_t hand_base, int row0,
                                                int count, int max_rows, int tile0, int wave, int lane) {
    const int m = lane & 31, h = lane >> 5;
    // back to true units with the row's activation scale and the layer's weight scale
    static_for<NT>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(row0 + 32 * (i / NU) + (r & 3) + 8 * (r >> 2) + 4 * h, max_rows - 1);
            acc[i][r] *= a_inv_scale[row] * inv_sw;
        }
    });
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(fin.handoff + hand_base, 0, NW * NT * 1024 * 4, 0x00027000);
    const int hoff = (wave * NT * 4 * 64 + lane) * 16;                 // + (i * 4 + j) * 1024 bytes
    if ((ticket & 1u) == 0u) {                                          // first of the pair: hand the tile over and leave
        static_for<NT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = __float_as_uint(acc[i][4 * j + c]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rh, hoff + (i * 4 + j) * 1024, 0, 16);    // 16 = sc1: write-through
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // every storing wave drains its stores ...
        __syncthreads();
        if (threadIdx.x == 0)                                           // ... before one lane raises the flag
            __hip_atomic_store(fin.flag + pair, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x == 0) {
        bool seen = false;
        for (int spin = 0; spin < (1 << 20); ++spin) {
            if (__hip_atomic_load(fin.flag + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ticket) { seen = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        if (!seen) atomicAdd(fin.err, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // x = ReLU((first half + second half) + bias); additions commute, so which half arrived first does not matter
    float bias[NU];
    static_for<NU>([&](auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        bias[u] = fin.fc_bias[(tile0 + u) * 32 + m];
    });
    static_for<NT>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rh, hoff + (i * 4 + j) * 1024, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][4 * j + c] = fmaxf((acc[i][4 * j + c] + __uint_as_float(v[c])) + bias[i % NU], 0.f);
        }
    });
    if (fin.hidden) {
        static_for<NT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 32 * (i / NU) + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < count) fin.hidden[(size_t)row * FCS_N + (tile0 + i % NU) * 32 + m] = acc[i][r];
            }
        });
    }
    // heads: per (row, head, column tile) the tree sum over the tile's 32 columns of x * head weight
    const int rsel = ((m >> 4) & 1) * 8 + ((m >> 3) & 1) * 4 + ((m >> 2) & 1) * 2 + ((m >> 1) & 1);   // row this lane ends up with
    const int rrow = (rsel & 3) + 8 * (rsel >> 2) + 4 * h;
    for (int o = fin.heads_all ? 0 : 7; o < 8; ++o) {
        float hw[NU];
        static_for<NU>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            hw[u] = fin.heads_w[o * FCS_N + (tile0 + u) * 32 + m];
        });
        static_for<NT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = acc[i][r] * hw[i % NU];
            const float sum = tile_column_sum(p, m);
            const int row = row0 + 32 * (i / NU) + rrow;
            if (!(m & 1) && row < count) fin.q[(size_t)row * 256 + o * 32 + tile0 + i % NU] = sum;
        });
    }
}

// ABL (timing experiments only, 0 in the product): 1 no activation loads, 2 no weight loads, 4 no LDS
// staging writes, 8 no fragment reads.
template <int MTR, int BR, int ABL = 0>
__device__ __forceinline__ void fc_split_tile(const u32x4 *__restrict__ a, const float *__restrict__ a_inv_scale,
                                              const u32x4 *__restrict__ wph, const u32x4 *__restrict__ wpl,
                                              const SplitScales *__restrict__ sc, const FcFinish &fin, int count,
                                              int max_rows, unsigned char *lds, int mbx) {
    const int bid = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 31, h = lane >> 5;
    // a group of blocks = 8 XCDs x (mbx row blocks x 4 column blocks x 2 K halves); mbx = 4 (256 blocks) or 2 (128)
    const int x = bid & 7, i = bid >> 3;
    const int per = 8 * mbx;
    const int grp = i / per, ii = i - grp * per;
    // XCDs 0-3 (blocks with blockIdx % 8 < 4, as blocks are dealt out today: a matter of traffic only) take column
    // blocks 0-1, XCDs 4-7 column blocks 2-3; inside a half, XCD x & 3 takes a quarter of the group's row blocks.
    // An XCD's L2 then fetches HALF the weight matrix, every byte serving 2*mbx row blocks, and each activation row
    // is fetched by two XCDs: for 2048 rows 184 + 189 MB instead of the 92 + 377 MB of "one XCD, all columns".
    const int q = 2 * mbx;                                   // row blocks per XCD
    const int mb = grp * 8 * mbx + (x & 3) * q + ii % q;
    const int nb = (x >> 2) * 2 + ((ii / q) & 1), ksp = ii / (2 * q);
    const int row0 = mb * (32 * MTR);
    if (row0 >= count) return;
    NNB_SP_STAMP(0) NNB_SP_WALL(5)

    constexpr int KHALF = FCS_K / FCS_SPLIT;                 // 5760
    constexpr int NSLAB = KHALF / FCS_BK;                    // 90 slabs of 64 k = 4 k-steps each
    constexpr int NKS = NSLAB * 4;
    constexpr int NCH = 2 * MTR;                             // staging chunks per thread and slab (16 rows each)
    static_assert(NSLAB % 6 == 0 && 24 % BR == 0, "the slab loop is unrolled by 6 (3 LDS buffers x 2 register sets)");
    const int slab0 = ksp * NSLAB;

    // Addresses are a uniform base (scalar registers, advanced per slab / k-step) plus a per-lane 32-bit offset
    // that never changes.  Staging: chunk C of a thread is row 16C + tid/16, piece tid%16 (0-7 hi, 8-15 lo).
    const int prow = tid >> 4, pq = tid & 15;
    // (buffer loads: resource = this block's rows - the base is advanced to row0 in 64-bit arithmetic, so the 32-bit
    // offsets below stay inside one row block however many rows the array has -, scalar offset, per-lane offset)
    const int rows_here = min(max_rows - row0, 32 * MTR);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(a) + (size_t)row0 * FCS_ROW_BYTES), 0,
        rows_here * FCS_ROW_BYTES, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wph), 0, FCS_K * FCS_N * 2, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wpl), 0, FCS_K * FCS_N * 2, 0x00027000);
    unsigned aoff[NCH];
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        const int r = min(row0 + 16 * c + prow, max_rows - 1) - row0;
        aoff[c] = (unsigned)r * FCS_ROW_BYTES + (pq & 7) * 16 + (pq >> 3) * 256;
    });
    unsigned char *sdst = lds + prow * FCS_LROW + pq * 16;                   // + c * 16 rows, + buffer
    auto slab_bytes = [&](int s) __attribute__((always_inline)) { const int sg = slab0 + s; return (sg >> 1) * 512 + (sg & 1) * 128; };
    auto load_a = [&](int s, unsigned c_off) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(ra, (int)c_off, slab_bytes(s), 0);
    };

    u32x4 stg[2][NCH];
    // prologue: slabs 0 and 1 into LDS, slabs 2 and 3 in flight in the two register sets
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        stg[0][c] = load_a(0, aoff[c]); stg[1][c] = load_a(1, aoff[c]);
    });
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        *reinterpret_cast<u32x4 *>(sdst + c * 16 * FCS_LROW) = stg[0][c];
        *reinterpret_cast<u32x4 *>(sdst + FCS_LBUF + c * 16 * FCS_LROW) = stg[1][c];
    });
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        stg[0][c] = load_a(2, aoff[c]); stg[1][c] = load_a(3, aoff[c]);
    });
    __syncthreads();

    f32x16 acc[2 * MTR];                                     // tile 2t + u: row tile t, column tile u
    static_for<2 * MTR>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    });

    // weight fragment of (k-step, column tile u): ((kstep*2 + h)*1024 + n) * 16 bytes, n = 256 nb + 64 w + 32 u + m
    constexpr int WKS = 2 * FCS_N * 16;                      // bytes per k-step
    const int wbase = ((ksp * KHALF / 16) * 2 * FCS_N + nb * 256) * 16;
    const int woff = (h * FCS_N + w * 64 + m) * 16;
    const __amdgpu_buffer_rsrc_t &wbh = rwh, &wbl = rwl;
    auto load_w = [&](const __amdgpu_buffer_rsrc_t &r, int kstep, int u) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, woff + u * 512, wbase + kstep * WKS, 0);
    };
    u32x4 bh[BR][2], bl[BR][2];
    static_for<BR - 1>([&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        bh[j][0] = load_w(wbh, j, 0); bh[j][1] = load_w(wbh, j, 1);
        bl[j][0] = load_w(wbl, j, 0); bl[j][1] = load_w(wbl, j, 1);
    });
    // fragment of (buffer, row tile t, k-step ks): row 32t + m, piece 2*ks + h (hi), 8 + 2*ks + h (lo)
    const unsigned char *fbase = lds + m * FCS_LROW + h * 16;
    auto frag = [&](int buf, int t, int ks, int lo) __attribute__((always_inline)) {
        return *reinterpret_cast<const u32x4 *>(fbase + buf * FCS_LBUF + t * 32 * FCS_LROW + ks * 32 + lo * 128);
    };
    // With 3 or 4 row tiles a tile's fragments are read again right after its last MFMA of the k-step (the
    // other tiles' MFMAs cover the read); with fewer there is not enough work in between and two sets alternate.
    constexpr int FSETS = MTR >= 3 ? 1 : 2;
    u32x4 fh[FSETS][MTR], fl[FSETS][MTR];
    static_for<MTR>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        fh[0][t] = frag(0, t, 0, 0);
        fl[0][t] = frag(0, t, 0, 1);
    });

    for (int s6 = 0; s6 < NSLAB; s6 += 6) {
        static_for<6>([&](auto Uc) __attribute__((always_inline)) {
            constexpr int U = decltype(Uc)::value;
            const int s = s6 + U;
            constexpr int buf = U % 3, set = U & 1;
            static_for<4>([&](auto Kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(Kc)::value;
                constexpr int J = U * 4 + ks;                    // k-step inside the 24-step body
                constexpr int cur = FSETS == 2 ? (J & 1) : 0, nxt = FSETS == 2 ? (cur ^ 1) : 0;
                constexpr int slot = J % BR, fill = (J + BR - 1) % BR;
                int knext = s * 4 + ks + BR - 1;
                knext = knext < NKS ? knext : NKS - 1;
                constexpr int nbuf = ks < 3 ? buf : (U + 1) % 3; // where the next k-step's fragments live
                constexpr int nks = (ks + 1) & 3;
                static_for<MTR>([&](auto T) __attribute__((always_inline)) {
                    constexpr int t = decltype(T)::value;
                    static_for<2>([&](auto Uu) __attribute__((always_inline)) {
                        constexpr int u = decltype(Uu)::value;
                        const half8 xh = as_half8(fh[cur][t]), xl = as_half8(fl[cur][t]);
                        const half8 wh = as_half8(bh[slot][u]), wl = as_half8(bl[slot][u]);
                        acc[2 * t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, acc[2 * t + u], 0, 0, 0);
                        acc[2 * t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc[2 * t + u], 0, 0, 0);
                        acc[2 * t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc[2 * t + u], 0, 0, 0);
                        // two memory instructions behind every MFMA triple
                        if constexpr (t == 0 && !(ABL & 2)) {    // weights of k-step J + BR - 1
                            bh[fill][u] = load_w(wbh, knext, u);
                            bl[fill][u] = load_w(wbl, knext, u);
                        }
                        if constexpr (u == 1 && !(ABL & 8)) {    // fragments of the next k-step, row tile t
                            fh[nxt][t] = frag(nbuf, t, nks, 0);     // (after the last k-step: read, never used)
                            fl[nxt][t] = frag(nbuf, t, nks, 1);
                        }
                        if constexpr (u == 0 && ks < 2 && !(ABL & 4)) {   // slab s+2: registers -> LDS
                            constexpr int c = ks * MTR + t;
                            // (past the last slab this rewrites a buffer nobody reads any more)
                            *reinterpret_cast<u32x4 *>(sdst + ((U + 2) % 3) * FCS_LBUF + c * 16 * FCS_LROW) = stg[set][c];
                        }
                        if constexpr (u == 0 && ks >= 2 && !(ABL & 1)) {  // slab s+4: global -> registers
                            constexpr int c = (ks - 2) * MTR + t;
                            stg[set][c] = load_a(s + 4 < NSLAB ? s + 4 : NSLAB - 1, aoff[c]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            });
            __syncthreads();
        });
    }

    // the two K halves of this tile meet (fc_split_finish); the ticket goes to every wave through LDS, which the
    // loop's closing barrier has released
    const int pair = mb * 4 + nb;
    if (tid == 0)
        *reinterpret_cast<unsigned *>(lds) = __hip_atomic_fetch_add(fin.ticket + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<const unsigned *>(lds);
    fc_split_finish<2 * MTR, 2, 4>(acc, fin, a_inv_scale, sc->inv_swf, ticket, pair,
                                   (size_t)row0 * FCS_N + (size_t)nb * (32 * MTR * 256), row0, count, max_rows,
                                   nb * 8 + w * 2, w, lane);
    NNB_SP_STAMP(3) NNB_SP_WALL(6)
}

template <int BR, int ABL = 0>
__global__ __launch_bounds__(256, 1) void fc_split_kernel(const u32x4 *__restrict__ a,          // [rows][90][32 chunks]
                                                          const float *__restrict__ a_inv_scale, // [rows]
                                                          const u32x4 *__restrict__ wph, const u32x4 *__restrict__ wpl,
                                                          const SplitScales *__restrict__ sc, FcFinish fin,
                                                          const int32_t *__restrict__ count_ptr, int n_fixed,
                                                          int max_rows, int shared_chip, unsigned long long *__restrict__ clk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * FCS_LBUF];    // three slab buffers
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    const int count = batch_count_sp(count_ptr, n_fixed);
    int mblocks = gridDim.x >> 3;                            // row blocks the grid provides (8 column/K blocks each)
    int mbx = 4;
    // With a companion forward pass on the chip and 1025..2048 rows, the first 128 blocks take 16 row blocks of 128
    // rows (where all 256 would take 32 of 64 rows): the companion's kernels fill the other CUs, and every weight
    // byte a block fetches serves twice the rows.
    if (shared_chip && gridDim.x == 256 && count > 1024 && count <= 2048) {
        if (blockIdx.x >= 128) return;
        mblocks = 16;
        mbx = 2;
    }
    const int tiles = (count + 32 * mblocks - 1) / (32 * mblocks);
    if (tiles <= 1) fc_split_tile<1, BR, ABL>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    else if (tiles == 2) fc_split_tile<2, BR, ABL>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    else if (tiles == 3) fc_split_tile<3, BR, ABL>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    else fc_split_tile<4, BR, ABL>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

// Small batches (at most kSplitSmallFc rows: single searches, small tournaments, the last games of a round) are
// latency-bound, and fc_split_kernel would give a handful of rows to 8 blocks.  Here a block is ONE wave that
// owns 32 rows x 32 columns x one half of K, so 32 rows spread over 64 CUs.  Fragments go memory -> registers
// (activations: 16 bytes per lane straight from the [row][cell][hi | lo] layout; weights as in fc_split_kernel)
// through a ring DEPTH k-steps deep.  Every output sums the same products in the same order as in
// fc_split_kernel (k-steps ascending inside a K half; hi*lo, lo*hi, hi*hi per k-step), so the results are
// the same bit for bit whatever the batch size.
#ifndef NNB_FC_SMALL_DEPTH
#define NNB_FC_SMALL_DEPTH 12
#endif
template <int DEPTH>
__global__ __launch_bounds__(64) void fc_split_small_kernel(const u32x4 *__restrict__ a, const float *__restrict__ a_inv_scale,
                                                            const u32x4 *__restrict__ wph, const u32x4 *__restrict__ wpl,
                                                            const SplitScales *__restrict__ sc, FcFinish fin,
                                                            const int32_t *__restrict__ count_ptr, int n_fixed, int max_rows,
                                                            unsigned long long *__restrict__ clk) {
    const int count = batch_count_sp(count_ptr, n_fixed);
    const int bid = blockIdx.x, rt = bid >> 6, ct = (bid >> 1) & 31, ksp = bid & 1;
    const int row0 = rt * 32;
    if (row0 >= count) return;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    constexpr int KHALF = FCS_K / FCS_SPLIT, NKS = KHALF / 16;           // 360 k-steps of 16
    // the resource starts at this block's first row (64-bit arithmetic), so 32-bit offsets stay inside 32 rows
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(a) + (size_t)row0 * FCS_ROW_BYTES), 0,
        min(max_rows - row0, 32) * FCS_ROW_BYTES, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wph), 0, FCS_K * FCS_N * 2, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wpl), 0, FCS_K * FCS_N * 2, 0x00027000);
    // activation fragment of k-step K (global index): 64 k per 256-byte half cell (128 B of hi at +0 / +128, lo 256 B
    // further): byte (K >> 3) * 512 + ((K >> 2) & 1) * 128 + (K & 3) * 32 + h * 16 of the row
    const int arow = min(row0 + m, max_rows - 1) - row0;
    const int aoff = arow * FCS_ROW_BYTES + h * 16;
    auto a_bytes = [&](int K) __attribute__((always_inline)) { return (K >> 3) * 512 + ((K >> 2) & 1) * 128 + (K & 3) * 32; };
    // weight fragment of k-step K: ((K * 2 + h) * 1024 + n) * 16 bytes, n = 32 ct + m
    const int woff = (h * FCS_N + ct * 32 + m) * 16;
    constexpr int WKS = 2 * FCS_N * 16;
    const int k0 = ksp * NKS;
    u32x4 xh[DEPTH], xl[DEPTH], wh[DEPTH], wl[DEPTH];
    static_for<DEPTH - 1>([&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        xh[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff, a_bytes(k0 + j), 0);
        xl[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + 256, a_bytes(k0 + j), 0);
        wh[j] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, (k0 + j) * WKS, 0);
        wl[j] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, (k0 + j) * WKS, 0);
    });
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    static_assert(NKS % DEPTH == 0, "the k loop is unrolled by the ring depth");
    for (int kb = 0; kb < NKS; kb += DEPTH) {
        static_for<DEPTH>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int j = decltype(Jc)::value;
            constexpr int fill = (j + DEPTH - 1) % DEPTH;
            int kn = kb + j + DEPTH - 1;
            kn = k0 + (kn < NKS ? kn : NKS - 1);                         // past the end: read again, never used
            xh[fill] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff, a_bytes(kn), 0);
            xl[fill] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + 256, a_bytes(kn), 0);
            wh[fill] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, kn * WKS, 0);
            wl[fill] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, kn * WKS, 0);
            const half8 ah = as_half8(xh[j]), al = as_half8(xl[j]), bh = as_half8(wh[j]), bl = as_half8(wl[j]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // the two K halves of this tile meet (fc_split_finish); neighbouring blocks hold them
    const int pair = rt * 32 + ct;
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(fin.ticket + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    f32x16 tile[1] = {acc};
    fc_split_finish<1, 1, 1>(tile, fin, a_inv_scale, sc->inv_swf, ticket, pair, (size_t)pair * 1024, row0, count, max_rows, ct, 0, lane);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

// ---------------------------------------------------------------------------
// The same linear layer on v_mfma_f32_16x16x32_f16 (fc16_*).  Under the socket's power cap this instruction shape
// delivers more than 32x32x16 on changing data (half the accumulator traffic per FLOP: tools/mfma_peak.hip), and the
// layer's time follows the clock the cap allows.  Block tile, K split, LDS slabs, weight layout, XCD mapping, the
// meeting of the two K halves and the heads' sums are those of fc_split_kernel; what changes is the shape of a wave's
// tiles: its 64 columns are FOUR 16-column tiles, the block's rows 16-row tiles, a k-step is 32 k.  Lane l supplies row
// l % 16 (A) / column l % 16 (B) for k = 8 * (l / 16) .. + 7 - LDS piece 4 * ks + l / 16 of the slab (hi; + 8: lo),
// weight piece (4 * kstep + l / 16) * 1024 + n - and holds D[4 * (l / 16) + r][l % 16], r = 0..3.
// Per accumulator and k-step: hi*lo, lo*hi, hi*hi over 32 k-values, k-steps ascending - the order both forms
// (fc16_kernel, fc16_small_kernel) keep, so that their results agree bit for bit.
// ---------------------------------------------------------------------------

// Sum over the 32 columns of a 32-column group of x * head weight, for the four rows a lane holds: the group is two
// 16-column tiles (columns c and c ^ 16 sit in the same lane: the tree's first level is an addition in place), then
// lanes 8, 4, 2, 1 apart; each of the first two exchanges halves the rows a lane carries.  Same tree as
// tile_column_sum.  Returns the sum of row 2 * bit3(m) + bit2(m).
__device__ __forceinline__ float fc16_group_sum(const f32x4 &xa, const f32x4 &xb, float hwa, float hwb, int m) {
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = __fadd_rn(__fmul_rn(xa[r], hwa), __fmul_rn(xb[r], hwb));
    const bool b3 = m & 8, b2 = m & 4;
    float v2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) v2[i] = (b3 ? p[2 + i] : p[i]) + lane_xor<8>(b3 ? p[i] : p[2 + i]);
    float v1 = (b2 ? v2[1] : v2[0]) + lane_xor<4>(b2 ? v2[0] : v2[1]);
    v1 = v1 + lane_xor<2>(v1);
    return v1 + lane_xor<1>(v1);
}

// fc_split_finish for 16x16 tiles: acc[t][u] covers rows row0 + 16 t .. + 15 and columns col0 + 16 u .. + 15 (col0 a
// multiple of 32, CT even).  hand_base: float offset of the pair's hand-off area (NW * RT * CT * 256 floats).
template <int RT, int CT, int NW>
__device__ __forceinline__ void fc16_finish(f32x4 (&acc)[RT][CT], const FcFinish &fin, const float *__restrict__ a_inv_scale,
                                            float inv_sw, unsigned ticket, int pair, size_t hand_base, int row0,
                                            int count, int max_rows, int col0, int wave, int lane) {
    static_assert(CT % 2 == 0, "two 16-column tiles form a 32-column group of the heads' sums");
    const int m = lane & 15, g = lane >> 4;
    static_for<RT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + 16 * t + 4 * g + r, max_rows - 1);
            const float k = a_inv_scale[row] * inv_sw;
#pragma unroll
            for (int u = 0; u < CT; ++u) acc[t][u][r] *= k;
        }
    });
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(fin.handoff + hand_base, 0, NW * RT * CT * 256 * 4, 0x00027000);
    const int hoff = (wave * RT * CT * 64 + lane) * 16;                // + (t * CT + u) * 1024 bytes
    if ((ticket & 1u) == 0u) {                                          // first of the pair: hand the tile over and leave
        static_for<RT>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int u = 0; u < CT; ++u) {
                u32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = __float_as_uint(acc[t][u][c]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rh, hoff + (t * CT + u) * 1024, 0, 16);    // 16 = sc1: write-through
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(fin.flag + pair, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x == 0) {
        bool seen = false;
        for (int spin = 0; spin < (1 << 20); ++spin) {
            if (__hip_atomic_load(fin.flag + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ticket) { seen = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        if (!seen) atomicAdd(fin.err, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    float bias[CT];
#pragma unroll
    for (int u = 0; u < CT; ++u) bias[u] = fin.fc_bias[col0 + 16 * u + m];
    static_for<RT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rh, hoff + (t * CT + u) * 1024, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][u][c] = fmaxf((acc[t][u][c] + __uint_as_float(v[c])) + bias[u], 0.f);
        }
    });
    if (fin.hidden) {
        static_for<RT>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * t + 4 * g + r;
#pragma unroll
                for (int u = 0; u < CT; ++u)
                    if (row < count) fin.hidden[(size_t)row * FCS_N + col0 + 16 * u + m] = acc[t][u][r];
            }
        });
    }
    const int rkeep = ((m >> 3) & 1) * 2 + ((m >> 2) & 1);               // the row this lane ends up with
    for (int o = fin.heads_all ? 0 : 7; o < 8; ++o) {
        float hw[CT];
#pragma unroll
        for (int u = 0; u < CT; ++u) hw[u] = fin.heads_w[o * FCS_N + col0 + 16 * u + m];
        static_for<RT>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int v = 0; v < CT / 2; ++v) {
                const float sum = fc16_group_sum(acc[t][2 * v], acc[t][2 * v + 1], hw[2 * v], hw[2 * v + 1], m);
                const int row = row0 + 16 * t + 4 * g + rkeep;
                if (!(m & 3) && row < count) fin.q[(size_t)row * 256 + o * 32 + col0 / 32 + v] = sum;
            }
        });
    }
}

// One matrix instruction with the accumulator tied to an accumulation register.  (Through the builtin the register
// allocator moves the 32 four-register accumulators of fc16_tile around between instructions: 405 copies per 1152
// instructions.)  The operands come from memory and LDS loads, whose waits the compiler places for the statement's
// inputs; a dependent instruction on the same accumulator follows at a distance of CT instructions.
#define NNB_FC16_MFMA(A, B, C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B))

template <int MTR, int BR>
__device__ __forceinline__ void fc16_tile(const u32x4 *__restrict__ a, const float *__restrict__ a_inv_scale,
                                          const u32x4 *__restrict__ wph, const u32x4 *__restrict__ wpl,
                                          const SplitScales *__restrict__ sc, const FcFinish &fin, int count,
                                          int max_rows, unsigned char *lds, int mbx) {
    const int bid = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, g = lane >> 4;
    // the mapping of blocks to (row block, column block, K half) of fc_split_tile
    const int x = bid & 7, i = bid >> 3;
    const int per = 8 * mbx;
    const int grp = i / per, ii = i - grp * per;
    const int q = 2 * mbx;
    const int mb = grp * 8 * mbx + (x & 3) * q + ii % q;
    const int nb = (x >> 2) * 2 + ((ii / q) & 1), ksp = ii / (2 * q);
    const int row0 = mb * (32 * MTR);
    if (row0 >= count) return;
    NNB_SP_STAMP(0) NNB_SP_WALL(5)

    constexpr int KHALF = FCS_K / FCS_SPLIT;                 // 5760
    constexpr int NSLAB = KHALF / FCS_BK;                    // 90 slabs of 64 k = 2 k-steps of 32
    constexpr int NKS = NSLAB * 2;
    constexpr int RT = 2 * MTR, CT = 4;                      // 16-row tiles, 16-column tiles
    constexpr int NCH = RT;                                  // staging chunks per thread and slab (16 rows each)
    static_assert(NSLAB % 6 == 0 && 12 % BR == 0, "the slab loop is unrolled by 6 (3 LDS buffers x 2 register sets)");
    const int slab0 = ksp * NSLAB;
    const int prow = tid >> 4, pq = tid & 15;
    const int rows_here = min(max_rows - row0, 32 * MTR);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(a) + (size_t)row0 * FCS_ROW_BYTES), 0,
        rows_here * FCS_ROW_BYTES, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wph), 0, FCS_K * FCS_N * 2, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wpl), 0, FCS_K * FCS_N * 2, 0x00027000);
    unsigned aoff[NCH];
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        const int r = min(row0 + 16 * c + prow, max_rows - 1) - row0;
        aoff[c] = (unsigned)r * FCS_ROW_BYTES + (pq & 7) * 16 + (pq >> 3) * 256;
    });
    unsigned char *sdst = lds + prow * FCS_LROW + pq * 16;                   // + c * 16 rows, + buffer
    auto slab_bytes = [&](int s) __attribute__((always_inline)) { const int sg = slab0 + s; return (sg >> 1) * 512 + (sg & 1) * 128; };
    auto load_a = [&](int s, unsigned c_off) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(ra, (int)c_off, slab_bytes(s), 0);
    };
    u32x4 stg[2][NCH];
    // prologue: slabs 0 and 1 into LDS, slabs 2 and 3 in flight in the two register sets
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        stg[0][c] = load_a(0, aoff[c]); stg[1][c] = load_a(1, aoff[c]);
    });
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        *reinterpret_cast<u32x4 *>(sdst + c * 16 * FCS_LROW) = stg[0][c];
        *reinterpret_cast<u32x4 *>(sdst + FCS_LBUF + c * 16 * FCS_LROW) = stg[1][c];
    });
    static_for<NCH>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        stg[0][c] = load_a(2, aoff[c]); stg[1][c] = load_a(3, aoff[c]);
    });
    __syncthreads();

    f32x4 acc[RT][CT];
    static_for<RT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    });
    // weight piece of (k-step of 32, column tile u): ((4 * kstep + g) * 1024 + n) * 16 bytes, n = 256 nb + 64 w + 16 u + m
    constexpr int WKS = 4 * FCS_N * 16;                      // bytes per k-step
    const int wbase = ((ksp * KHALF / 32) * 4 * FCS_N + nb * 256) * 16;
    const int woff = (g * FCS_N + w * 64 + m) * 16;
    auto load_w = [&](const __amdgpu_buffer_rsrc_t &r, int kstep, int u) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, woff + u * 256, wbase + kstep * WKS, 0);
    };
    u32x4 bh[BR][CT], bl[BR][CT];
    static_for<BR - 1>([&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
#pragma unroll
        for (int u = 0; u < CT; ++u) { bh[j][u] = load_w(rwh, j, u); bl[j][u] = load_w(rwl, j, u); }
    });
    // fragment of (buffer, row tile t, k-step ks of the slab): row 16t + m, piece 4*ks + g (hi), 8 + 4*ks + g (lo)
    const unsigned char *fbase = lds + m * FCS_LROW + g * 16;
    auto frag = [&](int buf, int t, int ks, int lo) __attribute__((always_inline)) {
        return *reinterpret_cast<const u32x4 *>(fbase + buf * FCS_LBUF + t * 16 * FCS_LROW + ks * 64 + lo * 128);
    };
    // fragments run AHEAD row tiles ahead of their use through a ring of AHEAD + 1 register sets
    constexpr int AHEAD = RT > 3 ? 3 : RT - 1, NSETS = AHEAD + 1;
    static_assert((RT * 12) % NSETS == 0, "the ring's period divides the unrolled body");
    u32x4 fh[NSETS], fl[NSETS];
    static_for<AHEAD>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        fh[t] = frag(0, t, 0, 0);
        fl[t] = frag(0, t, 0, 1);
    });

    for (int s6 = 0; s6 < NSLAB; s6 += 6) {
        static_for<6>([&](auto Uc) __attribute__((always_inline)) {
            constexpr int U = decltype(Uc)::value;
            const int s = s6 + U;
            constexpr int buf = U % 3, set = U & 1;
            static_for<2>([&](auto Kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(Kc)::value;
                constexpr int J = U * 2 + ks;                    // k-step inside the 12-step body
                constexpr int slot = J % BR, fill = (J + BR - 1) % BR;
                int knext = s * 2 + ks + BR - 1;
                knext = knext < NKS ? knext : NKS - 1;
                constexpr int nbuf = ks < 1 ? buf : (U + 1) % 3; // where the next k-step's fragments live
                constexpr int nks = (ks + 1) & 1;
                static_for<RT>([&](auto T) __attribute__((always_inline)) {
                    constexpr int t = decltype(T)::value;
                    constexpr int cur = (J * RT + t) % NSETS, nxt = (J * RT + t + AHEAD) % NSETS;
                    const u32x4 xh = fh[cur], xl = fl[cur];
                    // Per accumulator hi*lo, lo*hi, hi*hi; the four column tiles in between.  A wave issues one
                    // instruction every four cycles and a matrix instruction holds the pipe for sixteen, so at most
                    // three other instructions fit between two of them for free: the row tile's memory instructions
                    // are dealt out behind its three groups of four.
#pragma unroll
                    for (int u = 0; u < CT; ++u) NNB_FC16_MFMA(xh, bl[slot][u], acc[t][u]);
                    if constexpr (t + AHEAD < RT) {              // a later row tile's fragments of this k-step
                        fh[nxt] = frag(buf, t + AHEAD, ks, 0);
                        fl[nxt] = frag(buf, t + AHEAD, ks, 1);
                    } else {                                     // the first row tiles of the next k-step (after the last: read, never used)
                        fh[nxt] = frag(nbuf, t + AHEAD - RT, nks, 0);
                        fl[nxt] = frag(nbuf, t + AHEAD - RT, nks, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CT; ++u) NNB_FC16_MFMA(xl, bh[slot][u], acc[t][u]);
                    if constexpr (ks == 0) {                     // slab s+2: registers -> LDS (past the last slab this
                        *reinterpret_cast<u32x4 *>(sdst + ((U + 2) % 3) * FCS_LBUF + t * 16 * FCS_LROW) = stg[set][t];   // rewrites a buffer nobody reads any more)
                    } else {                                     // slab s+4: global -> registers
                        stg[set][t] = load_a(s + 4 < NSLAB ? s + 4 : NSLAB - 1, aoff[t]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CT; ++u) NNB_FC16_MFMA(xh, bh[slot][u], acc[t][u]);
                    // weights of k-step J + BR - 1 (the ring slot's last reader was the previous k-step), column tiles
                    // t, t + RT, ...
#pragma unroll
                    for (int u = t; u < CT; u += RT) {
                        bh[fill][u] = load_w(rwh, knext, u);
                        bl[fill][u] = load_w(rwl, knext, u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            __syncthreads();
        });
    }
    // the compiler does not see matrix instructions in the statements above: their last results are in place
    // 4 passes after issue, well inside these wait states
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    const int pair = mb * 4 + nb;
    if (tid == 0)
        *reinterpret_cast<unsigned *>(lds) = __hip_atomic_fetch_add(fin.ticket + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<const unsigned *>(lds);
    fc16_finish<RT, CT, 4>(acc, fin, a_inv_scale, sc->inv_swf, ticket, pair,
                           (size_t)row0 * FCS_N + (size_t)nb * (32 * MTR * 256), row0, count, max_rows,
                           nb * 256 + w * 64, w, lane);
    NNB_SP_STAMP(3) NNB_SP_WALL(6)
}

template <int BR>
__global__ __launch_bounds__(256, 1) void fc16_kernel(const u32x4 *__restrict__ a,          // [rows][90][32 chunks]
                                                      const float *__restrict__ a_inv_scale, // [rows]
                                                      const u32x4 *__restrict__ wph, const u32x4 *__restrict__ wpl,
                                                      const SplitScales *__restrict__ sc, FcFinish fin,
                                                      const int32_t *__restrict__ count_ptr, int n_fixed,
                                                      int max_rows, int shared_chip, unsigned long long *__restrict__ clk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * FCS_LBUF];    // three slab buffers
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    const int count = batch_count_sp(count_ptr, n_fixed);
    int mblocks = gridDim.x >> 3;                            // row blocks the grid provides (8 column/K blocks each)
    int mbx = 4;
#ifndef NNB_FC_HALF_GRID
#define NNB_FC_HALF_GRID 1
#endif
    if (NNB_FC_HALF_GRID && shared_chip && gridDim.x == 256 && count > 1024 && count <= 2048) {   // as fc_split_kernel
        if (blockIdx.x >= 128) return;
        mblocks = 16;
        mbx = 2;
    }
    const int tiles = (count + 32 * mblocks - 1) / (32 * mblocks);
    if (tiles <= 1) fc16_tile<1, BR>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    else if (tiles == 2) fc16_tile<2, BR>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    else if (tiles == 3) fc16_tile<3, BR>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    else fc16_tile<4, BR>(a, a_inv_scale, wph, wpl, sc, fin, count, max_rows, lds, mbx);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

// Small batches on the same tiles (cf. fc_split_small_kernel): a block is ONE wave that owns 32 rows x 32 columns x
// one half of K as 2 x 2 tiles of 16 x 16; fragments memory -> registers through a ring DEPTH k-steps (of 32) deep.
template <int DEPTH>
__global__ __launch_bounds__(64) void fc16_small_kernel(const u32x4 *__restrict__ a, const float *__restrict__ a_inv_scale,
                                                        const u32x4 *__restrict__ wph, const u32x4 *__restrict__ wpl,
                                                        const SplitScales *__restrict__ sc, FcFinish fin,
                                                        const int32_t *__restrict__ count_ptr, int n_fixed, int max_rows,
                                                        unsigned long long *__restrict__ clk) {
    const int count = batch_count_sp(count_ptr, n_fixed);
    const int bid = blockIdx.x, rt = bid >> 6, ct = (bid >> 1) & 31, ksp = bid & 1;
    const int row0 = rt * 32;
    if (row0 >= count) return;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    constexpr int KHALF = FCS_K / FCS_SPLIT, NKS = KHALF / 32;           // 180 k-steps of 32
    const int rows_here = min(max_rows - row0, 32);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(a) + (size_t)row0 * FCS_ROW_BYTES), 0,
        rows_here * FCS_ROW_BYTES, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wph), 0, FCS_K * FCS_N * 2, 0x00027000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(wpl), 0, FCS_K * FCS_N * 2, 0x00027000);
    // activation fragment of k-step K (global index, 32 k = a quarter of a cell's 128 channels), k-group g: byte
    // (K >> 2) * 512 + (K & 3) * 64 + g * 16 of the row (hi; the lo halves 256 bytes further)
    int aoff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) aoff[t] = (min(row0 + 16 * t + m, max_rows - 1) - row0) * FCS_ROW_BYTES + g * 16;
    auto a_bytes = [&](int K) __attribute__((always_inline)) { return (K >> 2) * 512 + (K & 3) * 64; };
    // weight piece of k-step K, column tile u: ((4 K + g) * 1024 + n) * 16 bytes, n = 32 ct + 16 u + m
    const int woff = (g * FCS_N + ct * 32 + m) * 16;
    constexpr int WKS = 4 * FCS_N * 16;
    const int k0 = ksp * NKS;
    u32x4 xh[DEPTH][2], xl[DEPTH][2], wh[DEPTH][2], wl[DEPTH][2];
    auto load_step = [&](int slot, int K) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xh[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[t], a_bytes(K), 0);
            xl[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[t] + 256, a_bytes(K), 0);
            wh[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff + t * 256, K * WKS, 0);
            wl[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff + t * 256, K * WKS, 0);
        }
    };
    static_for<DEPTH - 1>([&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        load_step(j, k0 + j);
    });
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_assert(NKS % DEPTH == 0, "the k loop is unrolled by the ring depth");
    for (int kb = 0; kb < NKS; kb += DEPTH) {
        static_for<DEPTH>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int j = decltype(Jc)::value;
            constexpr int fill = (j + DEPTH - 1) % DEPTH;
            int kn = kb + j + DEPTH - 1;
            kn = k0 + (kn < NKS ? kn : NKS - 1);                         // past the end: read again, never used
            load_step(fill, kn);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_half8(xh[j][t]), as_half8(wl[j][u]), acc[t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_half8(xl[j][t]), as_half8(wh[j][u]), acc[t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_half8(xh[j][t]), as_half8(wh[j][u]), acc[t][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const int pair = rt * 32 + ct;
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(fin.ticket + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    fc16_finish<2, 2, 1>(acc, fin, a_inv_scale, sc->inv_swf, ticket, pair, (size_t)pair * 1024, row0, count, max_rows, ct * 32, 0, lane);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

// The heads from the linear layer's per-tile partial sums (heads_math.h: heads_z_from_q), then log-softmax, tanh and
// the legal-move mask.  One wave per row.  The search itself does not launch this kernel (the tree step finishes its
// own leaf from the same sums); it serves nnb_net_forward and the checks.
__global__ __launch_bounds__(256) void heads_q_kernel(const float *__restrict__ q, const float *__restrict__ hb,
                                                      const uint8_t *__restrict__ legal, float *__restrict__ logp_out,
                                                      float *__restrict__ value_out, float *__restrict__ prior_out,
                                                      const int32_t *__restrict__ count_ptr, int n_fixed) {
    const int count = batch_count_sp(count_ptr, n_fixed);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave; row < count; row += nwaves) {
        float z[8];
        heads_z_from_q(q + (size_t)row * 256, hb, lane, z);
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

// Activations of the split path back to float32 [n][cells][128] (tests and debugging only).
__global__ void split_to_f32_kernel(const _Float16 *__restrict__ act, const float *__restrict__ inv_scale,
                                    int cells, int n, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * cells * 128) return;
    const size_t pos = i >> 7;
    const int ch = (int)(i & 127);
    const size_t board = pos / cells;
    out[i] = ((float)act[pos * 256 + ch] + (float)act[pos * 256 + 128 + ch]) * inv_scale[board];
}

// ---------------------------------------------------------------------------
// Host launchers
// ---------------------------------------------------------------------------
static inline int cdiv_sp(int a, int b) { return (a + b - 1) / b; }

// Up to this many rows the linear layer runs its latency-oriented form (one wave per block); the convolution kernel
// is one board per block at every batch size.  Whole forward passes: tools/net_latency.py.
#ifndef NNB_FC_SMALL_ROWS
#define NNB_FC_SMALL_ROWS 256
#endif
#ifndef NNB_FC16
#define NNB_FC16 1          // the linear layer on 16x16x32 tiles (fc16_kernel, fc16_small_kernel); 0: on 32x32x16 tiles
#endif                      // (fc_split_kernel, fc_split_small_kernel: kept for the A/B, tools/build_variants.py fc32)
constexpr int kSplitSmallFc = NNB_FC_SMALL_ROWS;
static_assert(2 * conv_board_lds_bytes() <= 160 * 1024, "two blocks of the convolution kernel share a CU");

hipError_t prepare_net_kernels_split() {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_board_split_kernel<4, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_board_lds_bytes());
    if (err != hipSuccess) return err;
    err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_board_split_kernel<4, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_board_lds_bytes());
    if (err != hipSuccess) return err;
    err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_spread_first_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_spread_lds_bytes());
    if (err != hipSuccess) return err;
    err = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_spread_second_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_spread_lds_bytes());
    if (err != hipSuccess) return err;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(conv_spread_second_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv_spread_lds_bytes());
}

// The convolutions of at most kSpreadMaxBoards boards, four blocks per board (three launches, see above).
static void launch_conv_spread(hipStream_t st, const NetWeights &w, const NetBuffers &b, const u64 *side, bool write_mid,
                               const int32_t *count_ptr, int n_max) {
    u32x4 *a2 = reinterpret_cast<u32x4 *>(b.act2), *a3 = reinterpret_cast<u32x4 *>(b.act3);
    conv_spread_first_kernel<<<4 * n_max, 256, conv_spread_lds_bytes(), st>>>(
        side, (const u32x4 *)w.conv0_wh, (const u32x4 *)w.conv0_wl, w.conv0_b, (const u32x4 *)w.conv2_wh,
        (const u32x4 *)w.conv2_wl, w.conv2_b, w.sp_scales, b.spread_f32, b.spread_max, count_ptr, n_max, b.clk);
    if (write_mid)
        conv_spread_second_kernel<true><<<4 * n_max, 256, conv_spread_lds_bytes(), st>>>(
            (const u32x4 *)w.conv4_wh, (const u32x4 *)w.conv4_wl, w.conv4_b, w.sp_scales, b.spread_f32, b.spread_max, a2,
            b.act2_inv, count_ptr, n_max);
    else
        conv_spread_second_kernel<false><<<4 * n_max, 256, conv_spread_lds_bytes(), st>>>(
            (const u32x4 *)w.conv4_wh, (const u32x4 *)w.conv4_wl, w.conv4_b, w.sp_scales, b.spread_f32, b.spread_max, nullptr,
            nullptr, count_ptr, n_max);
    conv_spread_pack_kernel<<<n_max, 256, 0, st>>>(b.spread_f32, b.spread_max, a3, b.act3_inv, count_ptr, n_max);
}

void launch_repack_split(hipStream_t st, const NetWeights &dst, const float *conv0_w, const float *conv0_b,
                         const float *conv2_w, const float *conv4_w, const float *fc_w) {
    unsigned *mx = reinterpret_cast<unsigned *>(dst.sp_maxbits);
    (void)hipMemsetAsync(mx, 0, 4 * sizeof(unsigned), st);
    absmax_kernel<<<4, 256, 0, st>>>(conv0_w, 128 * 32, mx + 0);
    absmax_kernel<<<64, 256, 0, st>>>(conv2_w, 128 * 128 * 16, mx + 1);
    absmax_kernel<<<64, 256, 0, st>>>(conv4_w, 128 * 128 * 16, mx + 2);
    absmax_kernel<<<1024, 256, 0, st>>>(fc_w, (size_t)1024 * 11520, mx + 3);
    split_scales_kernel<<<1, 64, 0, st>>>(mx, conv0_w, conv0_b, dst.sp_scales);
    repack_conv0_split_kernel<<<cdiv_sp(128 * 32, 256), 256, 0, st>>>(conv0_w, &dst.sp_scales->sw0, (_Float16 *)dst.conv0_wh,
                                                                      (_Float16 *)dst.conv0_wl);
    repack_conv_split_kernel<<<cdiv_sp(128 * 128 * 16, 256), 256, 0, st>>>(conv2_w, &dst.sp_scales->sw2, (_Float16 *)dst.conv2_wh,
                                                                           (_Float16 *)dst.conv2_wl);
    repack_conv_split_kernel<<<cdiv_sp(128 * 128 * 16, 256), 256, 0, st>>>(conv4_w, &dst.sp_scales->sw4, (_Float16 *)dst.conv4_wh,
                                                                           (_Float16 *)dst.conv4_wl);
    repack_fc_split_kernel<<<(1024 * 11520) / 256, 256, 0, st>>>(fc_w, &dst.sp_scales->swf, (_Float16 *)dst.fc_wh,
                                                                 (_Float16 *)dst.fc_wl);
}

void launch_net_stage_split(hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
                            const uint8_t *legal, float *logp_out, float *value_out, float *prior_out,
                            const int32_t *count_ptr, int n_max) {
    if (n_max <= 0) return;
    u32x4 *a2 = reinterpret_cast<u32x4 *>(b.act2), *a3 = reinterpret_cast<u32x4 *>(b.act3);
    switch (stage) {
    case NET_CONV_PAIR:         // all three convolutions in one launch; the activations between them stay in LDS
        if (n_max <= kSpreadMaxBoards && b.spread_f32) { launch_conv_spread(st, w, b, side, false, count_ptr, n_max); break; }
        conv_board_split_kernel<4, false><<<n_max, 256, conv_board_lds_bytes(), st>>>(
            side, (const u32x4 *)w.conv0_wh, (const u32x4 *)w.conv0_wl, w.conv0_b, (const u32x4 *)w.conv2_wh,
            (const u32x4 *)w.conv2_wl, w.conv2_b, (const u32x4 *)w.conv4_wh, (const u32x4 *)w.conv4_wl, w.conv4_b,
            w.sp_scales, nullptr, nullptr, a3, b.act3_inv, count_ptr, n_max, b.clk);
        break;
    case NET_CONV_PAIR_MID:     // the same, and the 8x9 activations are copied out (nnb_debug_activations)
        if (n_max <= kSpreadMaxBoards && b.spread_f32) { launch_conv_spread(st, w, b, side, true, count_ptr, n_max); break; }
        conv_board_split_kernel<4, true><<<n_max, 256, conv_board_lds_bytes(), st>>>(
            side, (const u32x4 *)w.conv0_wh, (const u32x4 *)w.conv0_wl, w.conv0_b, (const u32x4 *)w.conv2_wh,
            (const u32x4 *)w.conv2_wl, w.conv2_b, (const u32x4 *)w.conv4_wh, (const u32x4 *)w.conv4_wl, w.conv4_b,
            w.sp_scales, a2, b.act2_inv, a3, b.act3_inv, count_ptr, n_max, b.clk);
        break;
    case NET_FC: {
        const FcFinish fin = {w.fc_b, w.heads_w, b.fcq, b.write_hidden ? b.fc : nullptr, b.fchand, b.fcsync, b.fcsync + b.fc_pairs,
                              b.sync_err, b.heads_all};
        if (n_max <= kSplitSmallFc) {                            // one wave per 32 rows x 32 columns x K half
#if NNB_FC16
            fc16_small_kernel<NNB_FC_SMALL_DEPTH / 2><<<64 * cdiv_sp(n_max, 32), 64, 0, st>>>(
#else
            fc_split_small_kernel<NNB_FC_SMALL_DEPTH><<<64 * cdiv_sp(n_max, 32), 64, 0, st>>>(
#endif
                a3, b.act3_inv, (const u32x4 *)w.fc_wh, (const u32x4 *)w.fc_wl, w.sp_scales, fin, count_ptr, n_max,
                b.max_rows, b.clk ? b.clk + 4 : nullptr);
            break;
        }
        const int groups = cdiv_sp(cdiv_sp(n_max, 128), 32);       // 32 row blocks per group of 256 blocks
#if NNB_FC16
        fc16_kernel<FC16_BR><<<256 * groups, 256, 0, st>>>(
#else
        fc_split_kernel<FCS_BR><<<256 * groups, 256, 0, st>>>(
#endif
            a3, b.act3_inv, (const u32x4 *)w.fc_wh, (const u32x4 *)w.fc_wl, w.sp_scales, fin, count_ptr, n_max,
            b.max_rows, b.shared_chip, b.clk ? b.clk + 4 : nullptr);
        break;
    }
    case NET_HEADS: {   // needs the linear layer to have run with heads_all
        const int blocks = cdiv_sp(n_max, 4);                    // one wave per row
        heads_q_kernel<<<blocks < 512 ? blocks : 512, 256, 0, st>>>(b.fcq, w.heads_b, legal, logp_out, value_out, prior_out,
                                                                    count_ptr, n_max);
        break;
    }
    default: break;
    }
}

HeadsDev heads_input_split(const NetWeights &w, const NetBuffers &b) {
    return HeadsDev{b.fcq, nullptr, nullptr, nullptr, w.heads_w, w.heads_b, 0};
}

void launch_split_to_f32(hipStream_t st, const void *act, const float *inv_scale, int cells, int n, float *out) {
    if (n <= 0) return;
    const size_t total = (size_t)n * cells * 128;
    split_to_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const _Float16 *)act, inv_scale, cells, n, out);
}

}  // namespace nnb

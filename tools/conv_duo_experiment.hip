// An experiment kept beside the timing harness, not part of the library: the three convolutions for TWO boards per
// block.  It reproduces conv_board_split_kernel's outputs bit for bit and halves the weight bytes a board pulls from
// L2.  Alone on the chip it runs at a higher shader clock under the power cap and is about 10 % faster per launch;
// inside a self-play chain (between two launches of the linear layer) it takes the same time or longer, because it
// needs more cycles per board (255 k per pair against 2 x 123 k) and the chain's clock is not held down by the
// convolution's own weight stream.  Included by split_bench.hip after net_kernels_split.hip.
namespace nnb {
// ---------------------------------------------------------------------------
// The same convolutions for TWO boards per block (conv_duo_split_kernel, batches too large for latency to matter).
// What bounds conv_board_split_kernel is its weight stream - 2 MB of hi + lo weights from L2 per board - and a block
// may not grow beyond half a CU's LDS if two blocks are to share the CU.  Here the images hold PIXELS ONLY (a board's
// 8x9 input is 72 positions = 38 KB instead of 136 positions with their zero rows and columns), so two boards fit the
// same 76 KB: a lane whose tap falls outside its board reads one shared all-zero position instead (one address
// select per (tile, tap), from two 4-bit masks per lane and tile), nothing has to be zero-filled between the layers,
// and every weight fragment a wave loads serves the tiles of both boards.  The 144 / 180 output cells of the two
// boards are taken as ONE row-major sequence of 16-cell tiles (9 / 12 tiles; a tile may end one board and begin the
// other).  Each output sums the same products in the same order as in conv_board_split_kernel (a tap that is outside
// the board adds exact zeros there and here), so the two kernels agree bit for bit.
// ---------------------------------------------------------------------------
template <int IH, int IW>
struct DuoGeom {
    static constexpr int OH = IH + 1, OW = IW + 1, NC = OH * OW, NC2 = 2 * NC, NT = (NC2 + 15) / 16, IP = IH * IW;
    static constexpr int ZP = 2 * IP;                            // the all-zero position behind the two boards' pixels
    // kernel row kh of tile t reads at least one input row of some cell of the tile
    static constexpr bool live(int t, int kh) {
        for (int i = 0; i < 16; ++i) {
            int c = 16 * t + i;
            if (c >= NC2) break;
            const int r = (c % NC) / OW + kh - 2;
            if (r >= 0 && r < IH) return true;
        }
        return false;
    }
    static constexpr bool mixed(int t) { return 16 * t < NC && 16 * t + 15 >= NC; }      // cells of both boards
};

constexpr size_t conv_duo_lds_bytes() { return (size_t)(DuoGeom<8, 9>::ZP + 1) * SP_PB + 64; }
static_assert(DuoGeom<8, 9>::ZP + 1 >= 9 * 10 && DuoGeom<8, 9>::ZP >= DuoGeom<7, 8>::ZP + 1, "the region holds every image and one board's output");

// One tap (KH, kw) of one layer for the live tiles, in groups of TG tiles: the fragments of a group are read while the
// group before it multiplies; the weights of a k-step (2 channel tiles x hi, lo) are requested one k-step ahead.
// A lane's byte address for the tap is its pixel's, or the zero position's when the tap falls outside its board
// (tbase[t] + the tap's offset, chosen by two bits of the tile's mask byte: bits 0-3 kernel rows that stay on the
// board, bits 4-7 kernel columns); it is formed again for every k-step rather than kept for all tiles.
template <int IH, int IW, int NT, int KH, int TG>
__device__ __forceinline__ void conv_duo_tap(f32x4 (&acc)[NT][2], u32x4 (&bh)[2][2], u32x4 (&bl)[2][2],
                                             const unsigned char *smem, const int (&tbase)[NT],
                                             const unsigned (&masks)[(NT + 3) / 4], int zero_addr, int kw,
                                             const __amdgpu_buffer_rsrc_t &rwh, const __amdgpu_buffer_rsrc_t &rwl, int woff) {
    using G = DuoGeom<IH, IW>;
    constexpr int NG = (NT + TG - 1) / TG;                       // tile groups
    const int step0 = (KH * 4 + kw) * 4;
    const int tap_off = ((KH - 2) * IW + kw - 2) * SP_PB;
    u32x4 ah[2][TG], al[2][TG];
    auto load_group = [&](auto SetC, auto GrpC, auto KsC) __attribute__((always_inline)) {
        constexpr int set = decltype(SetC)::value, g = decltype(GrpC)::value, ks = decltype(KsC)::value;
        static_for<TG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, t = g * TG + i;
            if constexpr (t < NT) {
                if constexpr (G::live(t, KH)) {
                    const unsigned mk = masks[t / 4] >> ((t % 4) * 8);
                    const bool ok = (mk >> KH) & (mk >> (4 + kw)) & 1u;
                    const int a = ok ? tbase[t] + tap_off : zero_addr;
                    ah[set][i] = *reinterpret_cast<const u32x4 *>(smem + a + ks * 64);
                    al[set][i] = *reinterpret_cast<const u32x4 *>(smem + a + ks * 64 + SP_LO);
                }
            }
        });
    };
    load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    static_for<4 * NG>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value, ks = s / NG, g = s % NG, set = s & 1, slot = ks & 1;
#ifndef NNB_ABL_NO_CONV_WEIGHT_LOADS                              // (timing experiments only: tools/split_bench.hip)
        if constexpr (g == 0) {                                  // weights of the next k-step (past the layer's last: zeros)
            const int sn = (step0 + ks + 1) * 8192;
            bh[slot ^ 1][0] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, sn, 0);
            bh[slot ^ 1][1] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff + 256, sn, 0);
            bl[slot ^ 1][0] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, sn, 0);
            bl[slot ^ 1][1] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff + 256, sn, 0);
        }
#endif
        if constexpr (s + 1 < 4 * NG)
            load_group(std::integral_constant<int, set ^ 1>{}, std::integral_constant<int, (s + 1) % NG>{},
                       std::integral_constant<int, (s + 1) / NG>{});
        const half8 wh0 = as_half8(bh[slot][0]), wh1 = as_half8(bh[slot][1]);
        const half8 wl0 = as_half8(bl[slot][0]), wl1 = as_half8(bl[slot][1]);
        static_for<TG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value, t = g * TG + i;
            if constexpr (t < NT) {
                if constexpr (G::live(t, KH)) {
                    const half8 xh = as_half8(ah[set][i]), xl = as_half8(al[set][i]);
                    // per accumulator: hi*lo, lo*hi, hi*hi, in this order; the two channel tiles alternate
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl1, acc[t][1], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh1, acc[t][1], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh1, acc[t][1], 0, 0, 0);
                }
            }
        });
    });
}

// All four taps of kernel row KH.
template <int IH, int IW, int NT, int KH, int TG>
__device__ __forceinline__ void conv_duo_row(f32x4 (&acc)[NT][2], u32x4 (&bh)[2][2], u32x4 (&bl)[2][2],
                                             const unsigned char *smem, const int (&tbase)[NT], const unsigned (&masks)[(NT + 3) / 4],
                                             int zero_addr, const __amdgpu_buffer_rsrc_t &rwh, const __amdgpu_buffer_rsrc_t &rwl, int woff) {
    // (a real loop: the code of a block should stay resident in the instruction cache, 64 KB shared by two CUs)
#pragma unroll 1
    for (int kw = 0; kw < 4; ++kw)
        conv_duo_tap<IH, IW, NT, KH, TG>(acc, bh, bl, smem, tbase, masks, zero_addr, kw, rwh, rwl, woff);
}

template <int IH, int IW, int NT, int TG>
__device__ __forceinline__ void conv_duo_layer(f32x4 (&acc)[NT][2], const unsigned char *smem,
                                               const u32x4 *wph, const u32x4 *wpl, int lane, int wave) {
    using G = DuoGeom<IH, IW>;
    static_assert(NT == G::NT, "one accumulator pair per tile");
    const int m = lane & 15, kg = lane >> 4;
    int tbase[NT];
    unsigned masks[(NT + 3) / 4];
#pragma unroll
    for (int i = 0; i < (NT + 3) / 4; ++i) masks[i] = 0u;
    static_for<NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        int c = 16 * t + m;
        c = c < G::NC2 ? c : G::NC2 - 1;                         // rows past the last cell repeat it
        const int b = c >= G::NC ? 1 : 0, cc = c - b * G::NC, oh = cc / G::OW, ow = cc - oh * G::OW;
        tbase[t] = (b * G::IP + oh * IW + ow) * SP_PB + kg * 16;
        unsigned mk = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mk |= ((unsigned)(oh + k - 2) < (unsigned)IH ? 1u : 0u) << k;
            mk |= ((unsigned)(ow + k - 2) < (unsigned)IW ? 1u : 0u) << (4 + k);
        }
        masks[t / 4] |= mk << ((t % 4) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[t][0][r] = 0.f; acc[t][1][r] = 0.f; }
    });
    const int zero_addr = G::ZP * SP_PB + kg * 16;
    const __amdgpu_buffer_rsrc_t rwh = conv_weight_rsrc(wph), rwl = conv_weight_rsrc(wpl);
    const int woff = (kg * 128 + wave * 32 + m) * 16;           // + 256 bytes: the second channel tile
    u32x4 bh[2][2], bl[2][2];
    bh[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff, 0, 0);
    bh[0][1] = __builtin_amdgcn_raw_buffer_load_b128(rwh, woff + 256, 0, 0);
    bl[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff, 0, 0);
    bl[0][1] = __builtin_amdgcn_raw_buffer_load_b128(rwl, woff + 256, 0, 0);
    conv_duo_row<IH, IW, NT, 0, TG>(acc, bh, bl, smem, tbase, masks, zero_addr, rwh, rwl, woff);
    conv_duo_row<IH, IW, NT, 1, TG>(acc, bh, bl, smem, tbase, masks, zero_addr, rwh, rwl, woff);
    conv_duo_row<IH, IW, NT, 2, TG>(acc, bh, bl, smem, tbase, masks, zero_addr, rwh, rwl, woff);
    conv_duo_row<IH, IW, NT, 3, TG>(acc, bh, bl, smem, tbase, masks, zero_addr, rwh, rwl, woff);
}

// v = ReLU(acc * k[board] + bias) in place; this lane's maximum per board
template <int NT, int NC>
__device__ __forceinline__ void conv_duo_finish(f32x4 (&acc)[NT][2], const float (&k)[2], const float (&bv)[2], int kg, float (&mx)[2]) {
    mx[0] = mx[1] = 0.f;
    static_for<NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * t + 4 * kg + r;
            const bool second = 16 * t >= NC || (16 * t + 15 >= NC && c >= NC);
            const bool valid = 16 * t + 15 < 2 * NC || c < 2 * NC;
            const float kk = second ? k[1] : k[0];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v = fmaxf(__builtin_fmaf(acc[t][j][r], kk, bv[j]), 0.f);
                acc[t][j][r] = v;
                mx[0] = fmaxf(mx[0], valid && !second ? v : 0.f);
                mx[1] = fmaxf(mx[1], valid && second ? v : 0.f);
            }
        }
    });
}

// Quantised with its board's scale, hi / lo into LDS.  WHICH = 2: both boards, cell c of the two-board sequence goes
// to position c (the next layer's pixels-only image); WHICH = 0 / 1: only that board's cells, to position c - WHICH*NC
// (the dense output tile of one board).  Lane pairs pack two channels' halves into one 32-bit store (conv_board_store).
template <int NT, int NC, int WHICH>
__device__ __forceinline__ void conv_duo_store(const f32x4 (&acc)[NT][2], const float (&sc_out)[2], unsigned *ldsw, int wave, int lane) {
    const int m = lane & 15, kg = lane >> 4, odd = lane & 1;
    const unsigned psel = odd ? 0x03020706u : 0x05040100u;
    static_for<NT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        constexpr bool has0 = 16 * t < NC, has1 = 16 * t + 15 >= NC;
        if constexpr (WHICH == 2 || (WHICH == 0 && has0) || (WHICH == 1 && has1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * t + 4 * kg + r;
                const bool second = c >= NC;
                const float s = second ? sc_out[1] : sc_out[0];
                const int pos = WHICH == 2 ? c : c - WHICH * NC;
                const bool mine = c < 2 * NC && (WHICH == 2 || second == (WHICH == 1));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                    half2v pr;
                    const float x = acc[t][j][r] * s;
                    pr[0] = (_Float16)x;
                    pr[1] = (_Float16)(x - (float)pr[0]);
                    const unsigned own = __builtin_bit_cast(unsigned, pr);
                    const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // lane ^ 1
                    const int ch = wave * 32 + 16 * j + m;
                    if (mine) ldsw[pos * (SP_PB / 4) + (ch >> 1) + (odd ? SP_LO / 4 : 0)] = __builtin_amdgcn_perm(other, own, psel);
                }
            }
        }
    });
}

template <int TG, bool WRITE_MID>
__global__ __launch_bounds__(256, 2) void conv_duo_split_kernel(const u64 *__restrict__ side,
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
    using G1 = DuoGeom<7, 8>;
    using G2 = DuoGeom<8, 9>;
    constexpr int PB = SP_PB, IP1 = G1::IP, IP2 = G2::IP, OP2 = G2::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4 *lds = reinterpret_cast<u32x4 *>(smem_raw);
    float *s_max = reinterpret_cast<float *>(smem_raw + (G2::ZP + 1) * PB);   // [4 waves][2 boards]
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter(); clk[1] = wall_clock64(); }

    const int count = batch_count_sp(count_ptr, n_fixed);
    const int board0 = 2 * blockIdx.x;
    if (board0 >= count) return;
    const bool two = board0 + 1 < count;                         // an odd last board is computed twice, stored once
    const int tid = threadIdx.x;
    const int lane = tid & 63, nt = tid >> 6, m = lane & 31, h = lane >> 5;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    NNB_SP_STAMP(0) NNB_SP_WALL(5)

    // the two zero positions (behind the first and behind the second image's pixels)
    if (tid < 2 * SP_RS) lds[((tid < SP_RS ? G1::ZP : G2::ZP)) * SP_RS + tid % SP_RS] = zero4;
    // ---- leaf encode (AlphaZeroAgent.preprocess, agent_code.py:76-89) + 2->128 layer (network.py:29-30) into the first
    //      image: pixel p of board b at position b*56 + p (as conv_board_split_kernel, without the padding) ----
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
#pragma unroll 1
        for (int b = 0; b < 2; ++b) {
            const int board = board0 + (two ? b : 0);
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
                        _Float16 *dst = ldsh_in + (b * IP1 + p) * (PB / 2) + nt * 32 + m;
                        dst[0] = hi;
                        dst[128] = lo;
                    }
                }
            }
        }
    }
    __syncthreads();
    NNB_SP_STAMP(1)

    const int m16 = lane & 15, kg = lane >> 4;
    float k2[2];                                                 // 1 / (scale of a board's middle activations * weight scale)
    // ---- first 128->128 layer ----
    {
        f32x4 acc[G1::NT][2];
        conv_duo_layer<7, 8, G1::NT, TG>(acc, smem_raw, w2h, w2l, lane, nt);
        NNB_SP_STAMP(2)
        const float k1 = sc->inv_sa1 * sc->inv_sw2;
        const float kk[2] = {k1, k1};
        const float bv[2] = {bias2[nt * 32 + m16], bias2[nt * 32 + 16 + m16]};
        float mx[2];
        conv_duo_finish<G1::NT, G1::NC>(acc, kk, bv, kg, mx);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { mx[0] = fmaxf(mx[0], __shfl_xor(mx[0], off)); mx[1] = fmaxf(mx[1], __shfl_xor(mx[1], off)); }
        if (lane == 0) { s_max[2 * nt] = mx[0]; s_max[2 * nt + 1] = mx[1]; }
        __syncthreads();                                         // every wave is done with the first image
        float sout[2], sinv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
            sout[b] = pow2_scale(fmaxf(fmaxf(s_max[b], s_max[2 + b]), fmaxf(s_max[4 + b], s_max[6 + b])), sinv[b]);
        // the second image: cell c of the two-board sequence IS pixel c of the next layer's input
        conv_duo_store<G1::NT, G1::NC, 2>(acc, sout, reinterpret_cast<unsigned *>(smem_raw), nt, lane);
        const float inv_sw4 = sc->inv_sw4;
        k2[0] = sinv[0] * inv_sw4; k2[1] = sinv[1] * inv_sw4;
        if constexpr (WRITE_MID) {
            if (tid < (two ? 2 : 1)) mid_inv_scale[board0 + tid] = sinv[tid];
        }
        __syncthreads();                                         // the second image is complete (s_max may be reused)
        if constexpr (WRITE_MID) {
            u32x4 *dst = mid + (size_t)board0 * IP2 * 32;
            for (int e = tid; e < (two ? 2 : 1) * IP2 * 32; e += 256) dst[e] = lds[(e >> 5) * SP_RS + (e & 31)];
        }
    }
    NNB_SP_STAMP(4)

    // ---- second 128->128 layer ----
    f32x4 acc[G2::NT][2];
    conv_duo_layer<8, 9, G2::NT, TG>(acc, smem_raw, w4h, w4l, lane, nt);
    NNB_SP_STAMP(7)
    const float bv[2] = {bias4[nt * 32 + m16], bias4[nt * 32 + 16 + m16]};
    float mx[2];
    conv_duo_finish<G2::NT, G2::NC>(acc, k2, bv, kg, mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { mx[0] = fmaxf(mx[0], __shfl_xor(mx[0], off)); mx[1] = fmaxf(mx[1], __shfl_xor(mx[1], off)); }
    if (lane == 0) { s_max[2 * nt] = mx[0]; s_max[2 * nt + 1] = mx[1]; }
    __syncthreads();                                             // every wave is done with the second image
    float sout[2], sinv[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
        sout[b] = pow2_scale(fmaxf(fmaxf(s_max[b], s_max[2 + b]), fmaxf(s_max[4 + b], s_max[6 + b])), sinv[b]);
    // one board's output tile at a time through LDS, so that the stores to memory are whole rows
    conv_duo_store<G2::NT, G2::NC, 0>(acc, sout, reinterpret_cast<unsigned *>(smem_raw), nt, lane);
    __syncthreads();
    {
        u32x4 *dst = out + (size_t)board0 * OP2 * 32;
        for (int e = tid; e < OP2 * 32; e += 256) dst[e] = lds[(e >> 5) * SP_RS + (e & 31)];
    }
    __syncthreads();
    conv_duo_store<G2::NT, G2::NC, 1>(acc, sout, reinterpret_cast<unsigned *>(smem_raw), nt, lane);
    __syncthreads();
    if (two) {
        u32x4 *dst = out + (size_t)(board0 + 1) * OP2 * 32;
        for (int e = tid; e < OP2 * 32; e += 256) dst[e] = lds[(e >> 5) * SP_RS + (e & 31)];
    }
    if (tid < (two ? 2 : 1)) out_inv_scale[board0 + tid] = sinv[tid];
    NNB_SP_STAMP(3) NNB_SP_WALL(6)
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = __builtin_readcyclecounter(); clk[3] = wall_clock64(); }
}

}  // namespace nnb

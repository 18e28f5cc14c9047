// The prototype behind fc16_kernel (not part of the product; included by tools/split_bench.hip): the split linear layer's main loop on
// v_mfma_f32_16x16x32_f16 instead of 32x32x16, to time it.  Same block tile (128 rows x 256 columns x one K half),
// same LDS slab layout and weight layout, same number of fragment reads, weight loads and matrix cycles per k:
//   wave w owns columns 64w .. 64w+63 as FOUR 16-column tiles and the block's rows as EIGHT 16-row tiles;
//   a k-step is 32 k: lane l supplies row l % 16 (A) / column l % 16 (B) for k = 8 * (l / 16) .. + 7, i.e. LDS piece
//   4 * ks + l / 16 of the slab (hi) and 8 + that (lo), and weight piece (4 * kstep32 + l / 16) * 1024 + n.
// The epilogue only keeps the accumulators alive (a sum per lane to memory): the timing is that of the main loop.
namespace nnb {

// ABL (ablations, timing only): 1 no activation loads, 2 no weight loads, 4 no LDS staging writes, 8 no fragment
// reads, 16 no barrier per slab.
template <int BR, int ABL = 0>
__global__ __launch_bounds__(256, 1) void fc16_proto_kernel(const u32x4 *__restrict__ a, const u32x4 *__restrict__ wph,
                                                            const u32x4 *__restrict__ wpl, float *__restrict__ sink,
                                                            int count, int max_rows) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * FCS_LBUF];
    constexpr int MTR = 4;
    const int bid = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, g = lane >> 4;
    const int mbx = 4;
    const int x = bid & 7, i = bid >> 3;
    const int per = 8 * mbx;
    const int grp = i / per, ii = i - grp * per;
    const int q = 2 * mbx;
    const int mb = grp * 8 * mbx + (x & 3) * q + ii % q;
    const int nb = (x >> 2) * 2 + ((ii / q) & 1), ksp = ii / (2 * q);
    const int row0 = mb * (32 * MTR);
    if (row0 >= count) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_sp_clk[0] = __builtin_readcyclecounter(); g_sp_clk[5] = wall_clock64(); }

    constexpr int KHALF = FCS_K / FCS_SPLIT;
    constexpr int NSLAB = KHALF / FCS_BK;                    // 90 slabs of 64 k = 2 k-steps of 32
    constexpr int NKS = NSLAB * 2;
    constexpr int NCH = 2 * MTR;
    static_assert(NSLAB % 6 == 0 && 12 % BR == 0, "the slab loop is unrolled by 6");
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
    unsigned char *sdst = lds + prow * FCS_LROW + pq * 16;
    auto slab_bytes = [&](int s) __attribute__((always_inline)) { const int sg = slab0 + s; return (sg >> 1) * 512 + (sg & 1) * 128; };
    auto load_a = [&](int s, unsigned c_off) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(ra, (int)c_off, slab_bytes(s), 0);
    };
    u32x4 stg[2][NCH];
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

    constexpr int RT = 2 * MTR, CT = 4;                      // 16-row tiles, 16-column tiles
    f32x4 acc[RT][CT];
    static_for<RT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    });
    // weight piece of (k-step of 32, column tile u): ((4 * kstep + g) * 1024 + n) * 16 bytes, n = 256 nb + 64 w + 16 u + m
    constexpr int WKS = 4 * FCS_N * 16;                      // bytes per k-step of 32
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
    constexpr int AHEAD = 3, NSETS = AHEAD + 1;
    static_assert((RT * 12) % NSETS == 0 && AHEAD < RT, "the ring's period divides the unrolled body");
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
                constexpr int nbuf = ks < 1 ? buf : (U + 1) % 3;
                constexpr int nks = (ks + 1) & 1;
                static_for<RT>([&](auto T) __attribute__((always_inline)) {
                    constexpr int t = decltype(T)::value;
                    constexpr int cur = (J * RT + t) % NSETS, nxt = (J * RT + t + AHEAD) % NSETS;
                    if constexpr (!(ABL & 8)) {
                    if constexpr (t + AHEAD < RT) {              // a later row tile's fragments of this k-step
                        fh[nxt] = frag(buf, t + AHEAD, ks, 0);
                        fl[nxt] = frag(buf, t + AHEAD, ks, 1);
                    } else {                                     // the first row tiles of the next k-step (after the last: read, never used)
                        fh[nxt] = frag(nbuf, t + AHEAD - RT, nks, 0);
                        fl[nxt] = frag(nbuf, t + AHEAD - RT, nks, 1);
                    }
                    }
                    const u32x4 xh = fh[cur], xl = fl[cur];
                    // per accumulator hi*lo, lo*hi, hi*hi; the four column tiles in between, so that no instruction
                    // waits for the one before it
                    // (inline assembly with the accumulator tied to an accumulation register: through the builtin the
                    // register allocator moves the 32 accumulators of this loop around between instructions)
#define FC16_MFMA(A, B, C) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B))
#pragma unroll
                    for (int u = 0; u < CT; ++u) FC16_MFMA(xh, bl[slot][u], acc[t][u]);
#pragma unroll
                    for (int u = 0; u < CT; ++u) FC16_MFMA(xl, bh[slot][u], acc[t][u]);
#pragma unroll
                    for (int u = 0; u < CT; ++u) FC16_MFMA(xh, bh[slot][u], acc[t][u]);
                    // memory instructions behind every row tile's 12 MFMAs
                    if constexpr (t >= CT && !(ABL & 2)) {       // weights of k-step J + BR - 1, column tile t - CT (the
                        bh[fill][t - CT] = load_w(rwh, knext, t - CT);   // ring slot's last reader was the previous k-step)
                        bl[fill][t - CT] = load_w(rwl, knext, t - CT);
                    }
                    if constexpr (ks == 0) {                     // slab s+2: registers -> LDS
                        if constexpr (!(ABL & 4))
                            *reinterpret_cast<u32x4 *>(sdst + ((U + 2) % 3) * FCS_LBUF + t * 16 * FCS_LROW) = stg[set][t];
                    } else {                                     // slab s+4: global -> registers
                        if constexpr (!(ABL & 1)) stg[set][t] = load_a(s + 4 < NSLAB ? s + 4 : NSLAB - 1, aoff[t]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if constexpr (!(ABL & 16)) __syncthreads();
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last matrix instructions have left the pipe
    float sum = 0.f;
    static_for<RT>([&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int u = 0; u < CT; ++u) sum += acc[t][u][0] + acc[t][u][1] + acc[t][u][2] + acc[t][u][3];
    });
    sink[(size_t)bid * 256 + tid] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_sp_clk[3] = __builtin_readcyclecounter(); g_sp_clk[6] = wall_clock64(); }
}

}  // namespace nnb

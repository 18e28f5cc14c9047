// Tile geometry shared by the half-input network kernels (net_kernels_f16.hip: one half per value;
// net_kernels_split.hip: a hi + lo pair of halves per value).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace nnb {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant
// expression inside the body, so register arrays indexed with it never need an address.
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// 16 bytes as a native vector: register arrays of these are plain SSA values to the compiler
// (arrays of HIP's uint4 struct were seen to stay in memory)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

__device__ __forceinline__ half8 as_half8(uint4 v) {
    union { uint4 u; half8 h; } x;
    x.u = v;
    return x.h;
}

// MFMA row (lane & 31) -> output cell within the tile.  ds_read_b128 services lanes
// {0-3,12-15,20-27} and {4-11,16-19,28-31} (and the same +32) as two groups of 16; sigma
// gives the first group cells 0..15 and the second cells 16..31.
__host__ __device__ constexpr int sigma(int m) {
    return m < 4 ? m : m < 12 ? m + 12 : m < 16 ? m - 8 : m < 20 ? m + 8 : m < 28 ? m - 12 : m;
}

// Row-tile convolution image: positions between the first pixels of consecutive boards, chosen
// == OW (mod 16) so that the 16 cells one ds_read_b128 services together fall on 16 different
// bank groups even when they span two boards.
__host__ __device__ constexpr int conv_board_stride(int ih, int iw) {
    int s = ih * (iw + 2);
    while ((s & 15) != ((iw + 1) & 15)) ++s;
    return s;
}
// Output rows [rows_t0, rows_t1) are the ones whose kernel row kh reads at least one input row
// (4x4 kernel, padding 2); the others multiply nothing but padding and are skipped.
__host__ __device__ constexpr int rows_t0(int kh) { return kh < 2 ? 2 - kh : 0; }
__host__ __device__ constexpr int rows_t1(int ih, int kh) { return (ih + 2 - kh) < (ih + 1) ? (ih + 2 - kh) : (ih + 1); }

}  // namespace nnb

// Connect Four rules on 7x7-strided bitboards, shared by every kernel.
//
// Bit index = col*7 + row_from_bottom, rows 0..5 used, bit 6 of each column kept
// zero so shifted ANDs never wrap into a neighbouring column or row.
// Replaces the numpy board scans of nnbattle/game/connect_four_game.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c4 {

typedef unsigned long long u64;

constexpr u64 kBoardMask = 0xFDFBF7EFDFBFULL;   // 6 playable bits of each of 7 columns
constexpr u64 kTopMask   = 0x810204081020ULL;   // row_from_bottom == 5 of each column
constexpr u64 kBottomMask = 0x40810204081ULL;   // row_from_bottom == 0 of each column

// connect_four_game.py:160-192 - is any of the 69 windows of four filled by `b`
__host__ __device__ __forceinline__ bool has_four(u64 b) {
    u64 v = b & (b >> 1);    // vertical
    u64 h = b & (b >> 7);    // horizontal
    u64 d = b & (b >> 6);    // diagonal, down-right in the reference's row order
    u64 e = b & (b >> 8);    // diagonal, up-right
    return ((v & (v >> 2)) | (h & (h >> 14)) | (d & (d >> 12)) | (e & (e >> 16))) != 0;
}

// connect_four_game.py:194-200
__host__ __device__ __forceinline__ bool is_full(u64 occ) { return (occ & kBoardMask) == kBoardMask; }

// connect_four_game.py:72-78
__host__ __device__ __forceinline__ bool is_terminal(u64 b1, u64 b2) {
    return has_four(b1) || has_four(b2) || is_full(b1 | b2);
}

// connect_four_game.py:109-125 - bit c set iff the top cell of column c is empty
__host__ __device__ __forceinline__ unsigned legal_mask(u64 occ) {
    u64 t = (~occ & kTopMask) >> 5;              // one bit per column at position 7c
    // gather bits 0,7,14,...,42 into bits 0..6
    return (unsigned)(((t * 0x41041041040ULL) >> 42) & 0x7F);
}

// connect_four_game.py:46-49 - the cell a stone dropped in `col` lands in: the lowest
// empty cell of the column (the reference scans rows bottom-up for the first empty one).
// Caller has checked legal_mask.
__host__ __device__ __forceinline__ u64 drop_cell(u64 occ, int col) {
    u64 empty = ~occ & (0x3FULL << (col * 7));
    return empty & (0 - empty);
}

// connect_four_game.py:80-91 - absolute frame: colour 2 is tested first
__host__ __device__ __forceinline__ double reward(u64 b1, u64 b2) {
    return has_four(b2) ? 1.0 : (has_four(b1) ? -1.0 : 0.0);
}

// connect_four_game.py:280-296 - colour-1 frame, colour 1 tested first; 2 = not finished
__host__ __device__ __forceinline__ int result(u64 b1, u64 b2) {
    if (has_four(b1)) return 1;
    if (has_four(b2)) return -1;
    return is_full(b1 | b2) ? 0 : 2;
}

// 64-bit mix of the side boards behind the fixed evaluators (splitmix64 finaliser)
__host__ __device__ __forceinline__ u64 mix64(u64 me, u64 opp) {
    u64 x = me ^ (opp * 0x9E3779B97F4A7C15ULL);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

__host__ __device__ __forceinline__ double hash_value_q8(u64 me, u64 opp) {
    return ((double)(long long)(mix64(me, opp) >> 55) - 256.0) / 256.0;
}

__host__ __device__ __forceinline__ double hash_value_f32(u64 me, u64 opp) {
    u64 x = mix64(me, opp);
    double v = ((double)(x >> 40) / 8388608.0 - 1.0) / (double)(1u << ((x >> 37) & 7));
    return (double)(float)v;
}

}  // namespace c4

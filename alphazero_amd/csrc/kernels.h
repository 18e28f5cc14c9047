// Host-callable launchers of the engine's kernels (each defined next to its kernel).
#pragma once
#include <hip/hip_runtime.h>
#include "engine_layout.h"
#include "heads_math.h"

namespace nnb {

// tree_kernels.hip
void launch_game_ops(hipStream_t st, int n, const u64 *in, const int8_t *players, const int8_t *actions,
                     u64 *out, uint8_t *legal_out, uint8_t *status_out);
void launch_encode_planes(hipStream_t st, int n, const u64 *side, float *planes);
void launch_minimax(hipStream_t st, int n, const u64 *boards, const int8_t *movers, int depth, int8_t *col_out,
                    int32_t *score_out);
void launch_ucb_scores(hipStream_t st, int n, const double *value, const int32_t *visits,
                       const int32_t *parent_visits, double c_puct, const double *log_table, int log_n,
                       double *out);
void launch_tree_init(hipStream_t st, const EngineDev &e, const SearchDev &p, int make_noise, int selfplay);
// heads: when heads.fused, a wave with a pending leaf computes the network's heads for it from the
// linear layer's output (batch b of the leaf arrays reads heads.h[b]) instead of reading
// leaf_value / leaf_prior
struct HeadsPair {
    HeadsDev h[2];
    int32_t fused;
};
void launch_tree_step(hipStream_t st, const EngineDev &e, const SearchDev &p, int parity, const HeadsPair &heads,
                      const RoundDev *turnover = nullptr, const GameSpan *span = nullptr);
hipError_t prepare_tree_fused(size_t lds_bytes);
hipError_t prepare_tree_step();
// How the fused tree kernel uses its dynamic LDS for a search of `sims` simulations within `budget` bytes: the first
// `nodes` nodes of the pool (all 1 + 7*sims of them unless `split`), the log table too if `logt`.
struct FusedLds {
    size_t bytes;
    int nodes;
    int logt;
    bool split;
};
FusedLds fused_lds_plan(int sims, bool az, size_t budget);
void launch_tree_fused(hipStream_t st, const EngineDev &e, const SearchDev &p, const FusedLds &f);
void launch_readout(hipStream_t st, const EngineDev &e, const SearchDev &p, int32_t *visits_out,
                    int8_t *action_out, double *root_value_out, const RoundDev *round);
void launch_trajectory_samples(hipStream_t st, int n, const nnb_trajectory *rec, const int32_t *ply_offsets, float *planes,
                               float *probs, float *z);
void launch_selfplay_init(hipStream_t st, const EngineDev &e, const RoundDev &round);
void launch_turnover(hipStream_t st, const EngineDev &e, const SearchDev &p, const RoundDev &round, int make_noise,
                     const GameSpan *span = nullptr);
void launch_search_init(hipStream_t st, const EngineDev &e, const u64 *boards, const int8_t *players,
                        const int8_t *sides, int game_base, const float *root_noise);
void launch_dirichlet_noise(hipStream_t st, int n, u64 seed, const int32_t *game_ids, const int32_t *move_index,
                            const uint8_t *legal, double alpha, float *out);
void launch_stats_reduce(hipStream_t st, const EngineDev &e);
void launch_fixed_eval(hipStream_t st, const EngineDev &e, const SearchDev &p, int batch, int parity);

// Power-of-two scales of the split-operand path (net_kernels_split.hip), in device memory, written
// when the weights are loaded: s* puts the tensor's largest magnitude just under 2^14, inv_s* = 1/s*.
struct SplitScales {
    float sw0, inv_sw0;        // 2->128 layer weights
    float sa1, inv_sa1;        // that layer's output (static: bounded by its weights, the inputs being 0/1)
    float sw2, inv_sw2;        // first 128->128 layer weights
    float sw4, inv_sw4;        // second 128->128 layer weights
    float swf, inv_swf;        // linear layer weights
};

// net_kernels.hip
struct NetWeights {            // one resident weight set, kernel layouts
    float *conv0_w, *conv0_b;  // [128][2][4][4], [128]
    float *conv2_w, *conv2_b;  // repacked [16][16][2][128][4], [128]
    float *conv4_w, *conv4_b;
    float *fc_w, *fc_b;        // repacked [1440][2][1024][4], [1024]
    float *heads_w, *heads_b;  // [8][1024], [8]
    // half-precision-input path (net_kernels_f16.hip); null when the engine runs the f32 path
    void *conv2_wh, *conv4_wh; // half [16][8][2][128][8]
    void *fc_wh;               // half [720][2][1024][8]
    // split-operand path: the arrays above hold the hi halves of the scaled weights, these the lo halves
    void *conv2_wl, *conv4_wl, *fc_wl;
    void *conv0_wh, *conv0_wl; // half [128][32]
    SplitScales *sp_scales;
    void *sp_maxbits;          // 4 x uint32 scratch of the weight preparation
};
struct NetBuffers {            // activations, channels-last float32
    float *act1;               // [G][7][8][128]
    float *act2;               // [G][8][9][128]
    float *act3;               // [G][9][10][128]
    float *fc;                 // [G][1024]
    float *fcpart;             // [2][G][1024] split-K partial sums (f16 path)
    float *fc_b1, *fcpart_b1;  // the same two for the second leaf batch of a two-network arena
    float *act1_b1, *act2_b1, *act3_b1;   // and its convolution buffers (the two batches run on two streams)
    float *act2_inv, *act3_inv;           // split-operand path: 1 / scale of each board's activations, [G] each
    float *act2_inv_b1, *act3_inv_b1;
    // split-operand path, last step of the linear layer (fc_split_finish): per-column-tile partial sums of the heads
    // [G][8][32], the tile the first-arriving K half hands to its partner block ((G + 128) x 1024 floats), the
    // ticket and flag words of those pairs, a word that counts waits that ran out, and (batch 0 only) the hidden
    // activations ReLU(linear layer) [G][1024] for nnb_debug_activations
    float *fcq, *fcq_b1;
    float *fchand, *fchand_b1;
    uint32_t *fcsync, *fcsync_b1;
    int32_t *sync_err;
    // split-operand path, a handful of boards (at most kConvSpreadMaxBoards): four blocks share a board's convolutions
    // and meet through float32 activations [boards][(72 + 90) * 128] and two maximum words per board (zero between passes)
    float *spread_f32, *spread_f32_b1;
    uint32_t *spread_max, *spread_max_b1;
    int fc_pairs;              // ticket words per batch (the flag words follow them)
    int heads_all;             // 1: the linear layer's last step produces all 8 heads, 0: the value head only
    int write_hidden;          // 1: it also writes the hidden activations to `fc`
    unsigned long long *clk;   // non-null while profiling: [0..3] the convolution launch's, [4..7] the linear layer's
                               // (shader cycles, 100 MHz wall clock) at the start and at the end of block 0
    int max_rows;              // G
    int shared_chip;           // 1: another forward pass runs beside this one on a second stream (two chains of a
                               //    self-play round, the two networks of an arena): with 1025..2048 rows the linear
                               //    layer then runs half as many blocks with twice the rows each, since the companion
                               //    fills the other CUs and each weight byte fetched serves twice the rows
};
enum NetStage { NET_CONV0 = 0, NET_CONV2 = 1, NET_CONV4 = 2, NET_FC = 3, NET_HEADS = 4,
                // float32-faithful path only: NET_CONV0 + NET_CONV2 + NET_CONV4 as one launch (the activations between the
                // two 128->128 layers stay in LDS), and the same with those activations also copied out
                NET_CONV_PAIR = 5, NET_CONV_PAIR_MID = 6 };

constexpr int kConvSpreadMaxBoards = 128;
constexpr int kConvSpreadFloatsPerBoard = (72 + 90) * 128;

// Ticket / flag words of the linear layer's block pairs for G rows: the large-batch form pairs (row block of 32..128
// rows, 256-column block), the small-batch form (32-row tile, 32-column tile) for at most 256 rows.
constexpr int fc_sync_pairs(int max_rows) { return ((max_rows + 31) / 32 + 8) * 32; }

HeadsDev heads_input_f32(const NetWeights &w, const NetBuffers &b);
HeadsDev heads_input_split(const NetWeights &w, const NetBuffers &b);
HeadsDev heads_input_f16(const NetWeights &w, const NetBuffers &b);
hipError_t prepare_net_kernels();
void launch_repack(hipStream_t st, const NetWeights &dst, const float *conv0_w, const float *conv0_b,
                   const float *conv2_w, const float *conv2_b, const float *conv4_w, const float *conv4_b,
                   const float *fc_w, const float *fc_b, const float *policy_w, const float *policy_b,
                   const float *value_w, const float *value_b);
void launch_planes_to_bits(hipStream_t st, int n, const float *planes, u64 *side);
// One stage of the forward pass over `n_max` rows at most; the live row count is read
// from *count_ptr on the device when count_ptr != nullptr, else it is n_max.
void launch_net_stage(hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
                      const uint8_t *legal, float *logp_out, float *value_out, float *prior_out,
                      const int32_t *count_ptr, int n_max);

// net_kernels_f16.hip: same stages with half activations / weights on v_mfma_f32_32x32x16_f16
hipError_t prepare_net_kernels_f16();
void launch_repack_f16(hipStream_t st, const NetWeights &dst, const float *conv2_w, const float *conv4_w,
                       const float *fc_w);
void launch_net_stage_f16(hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
                          const uint8_t *legal, float *logp_out, float *value_out, float *prior_out,
                          const int32_t *count_ptr, int n_max);

// net_kernels_split.hip: same stages with every float32 operand carried as a hi + lo pair of halves
// (three MFMAs per product, float32-faithful)
hipError_t prepare_net_kernels_split();
void launch_repack_split(hipStream_t st, const NetWeights &dst, const float *conv0_w, const float *conv0_b,
                         const float *conv2_w, const float *conv4_w, const float *fc_w);
void launch_net_stage_split(hipStream_t st, int stage, const NetWeights &w, const NetBuffers &b, const u64 *side,
                            const uint8_t *legal, float *logp_out, float *value_out, float *prior_out,
                            const int32_t *count_ptr, int n_max);
void launch_split_to_f32(hipStream_t st, const void *act, const float *inv_scale, int cells, int n, float *out);

}  // namespace nnb

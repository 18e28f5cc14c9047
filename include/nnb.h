/*
 * nnb.h - C ABI of libnnb.so, the MI355X (gfx950) self-play engine behind the
 * nnbattle agent/game API.
 *
 * The reference (jahoops/alphazero, package `nnbattle`) has no FFI: its hot
 * path is Python.  Each entry point below names the reference code it replaces
 * (paths under the reference root).  The Python classes in alphazero_amd/ keep
 * the reference's names and signatures and call these functions through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success and a negative nnb_status on error;
 *     nnb_last_error() returns the message of the calling thread's last error;
 *   - all array arguments are DEVICE pointers owned by the caller (for example
 *     torch tensors' data_ptr()); the library never frees them;
 *   - work is enqueued on the caller's HIP stream (`stream`, a hipStream_t passed
 *     as void*; NULL = the default stream); functions documented as
 *     "synchronous" wait for the stream before returning;
 *   - one host thread per engine; calls on one engine must not overlap;
 *   - the engine owns no device memory of its own: it lives in a workspace the
 *     caller allocates (nnb_workspace_bytes) and hands to nnb_engine_create.
 *
 * Bitboards: a position is two uint64 words.  Bit index = col*7 + row_from_bottom
 * (row_from_bottom = 5 - the reference's row index; its row 0 is the top row,
 * nnbattle/game/connect_four_game.py:21,47).  Bit 6 of each 7-bit column is zero.
 * "Colour boards" are (colour-1 stones, colour-2 stones); "side boards" are
 * (stones of the side the network looks from, opponent stones), the bit form of
 * AlphaZeroAgent.preprocess (nnbattle/agents/alphazero/agent_code.py:76-89).
 */
#ifndef NNB_H
#define NNB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nnb_engine nnb_engine;

enum nnb_status {
    NNB_OK = 0,
    NNB_ERR_ARG = -1,       /* bad argument (null pointer, size out of range ...) */
    NNB_ERR_HIP = -2,       /* a HIP runtime call failed                          */
    NNB_ERR_STATE = -3,     /* call order violated (e.g. weights not loaded)      */
    NNB_ERR_WORKSPACE = -4  /* workspace too small or misaligned                  */
};

/* Search rule.  COMPAT reproduces the reference search bit for bit
 * (agent_code.py:144-234, mcts.py:25-44), including its quirks.  AZ is the PUCT
 * rule with priors, Dirichlet root noise, side-alternating descent and a
 * sign-flipped backup; it has no counterpart in the reference. */
enum nnb_mode { NNB_MODE_COMPAT = 0, NNB_MODE_AZ = 1 };

/* Leaf evaluator.  NET is the policy-value network (network.py:43-50).  The other
 * three are fixed functions of the leaf position used for bit-exact parity runs
 * and tree-only benchmarks: CONST = value 0 / uniform priors; HASH_Q8 and
 * HASH_F32 = a 64-bit mix of the side boards mapped to a multiple of 1/256 or to
 * a float32; ROLLOUT = uniform priors and a uniformly random playout. */
enum nnb_evaluator {
    NNB_EVAL_NET = 0,
    NNB_EVAL_CONST = 1,
    NNB_EVAL_HASH_Q8 = 2,
    NNB_EVAL_HASH_F32 = 3,
    NNB_EVAL_ROLLOUT = 4
};

/* nnb_search_params.flags */
enum nnb_flags {
    /* Run a fixed evaluator through the lock-step path (tree step kernel, leaf batch,
     * value buffer) that NET uses, instead of the fused one-launch kernel.  Same
     * results; exists so the lock-step machinery can be checked bit for bit. */
    NNB_FLAG_LOCKSTEP = 1,
    /* NET evaluator: run the network's heads as a kernel of their own after the linear layer
     * (values and priors of the leaf batch go through the engine's buffers) instead of inside
     * the tree step.  Same results; exists so the two forms can be checked against each other. */
    NNB_FLAG_SEPARATE_HEADS = 2,
    /* nnb_selfplay_round on the lock-step path: keep all games on a common move boundary
     * (S+1 tree steps, then one readout for everybody) instead of letting every game end its
     * move and start the next one as soon as its own simulations are done.  Same games; the
     * default keeps the leaf batch fuller when games differ. */
    NNB_FLAG_SYNC_MOVES = 4,
    /* Lock-step path: remember every leaf evaluation of a game (value, masked priors) in a per-game
     * hash table and reuse it when the same position comes up again in that game - transpositions
     * inside a search, and under PUCT the positions earlier moves already looked at.  A hit ends
     * its simulation on the spot like a finished position does.  The evaluators are functions of
     * the position, so the games are the same byte for byte; only the number of leaves sent to
     * the network drops.  Off by default (the reference evaluates every leaf, agent_code.py:195);
     * needs an engine created with eval_cache_slots > 0. */
    NNB_FLAG_EVAL_CACHE = 8,
    /* nnb_selfplay_round on the lock-step path with one network.  The slots can be split in two halves
     * that run their tree step -> forward pass chains on two streams, unsynchronised, so that one half's
     * tree step and the idle tail of its kernels hide under the other half's forward pass.  Same games
     * either way.  Default: two streams from 1024 slots on with NNB_NET_F32S / NNB_NET_F32, one stream with
     * NNB_NET_F16 (where it measured slower).  ONE_STREAM / TWO_STREAMS force the choice. */
    NNB_FLAG_ONE_STREAM = 16,
    NNB_FLAG_TWO_STREAMS = 32,
    /* Accepted and ignored (it once selected a two-launch form of the NNB_NET_F32S convolutions; one kernel now
     * serves every batch size, so a position's outputs cannot depend on the form that evaluated it). */
    NNB_FLAG_SEPARATE_CONVS = 64
};

/* How the network's three big layers multiply.  All accumulate in float32 and keep bias,
 * ReLU, heads, log-softmax and tanh in float32.
 *   NNB_NET_F16 : activations and weights rounded to IEEE half, v_mfma_f32_32x32x16_f16
 *                 (about 1e-5 absolute from the float32 forward on the default network);
 *   NNB_NET_F32 : float32 inputs, v_mfma_f32_32x32x2_f32 (exact f32 FMA chains, 1/16 the rate);
 *   NNB_NET_F32S: float32-faithful on the half-input matrix cores: every float32 operand is carried as
 *                 a scaled hi + lo pair of halves (2^-22 relative), three MFMAs per product, a third of
 *                 the NNB_NET_F16 rate; as close to the float32 forward as NNB_NET_F32 is. */
enum nnb_precision { NNB_NET_F16 = 0, NNB_NET_F32 = 1, NNB_NET_F32S = 2 };

typedef struct nnb_config {
    int32_t max_games;   /* concurrent games / root positions per call (>= 1)      */
    int32_t max_sims;    /* simulations per move; node pool = 1 + 7*max_sims nodes */
    int32_t device;      /* HIP device ordinal the workspace lives on              */
    int32_t n_weight_sets; /* resident network weight sets (1, or 2 for an arena)  */
    int32_t net_precision; /* nnb_precision: arithmetic of the conv / linear layers */
    int32_t eval_cache_slots; /* 0, or a power of two <= 65536: per-game table of leaf evaluations
                                 (64 B per slot) for NNB_FLAG_EVAL_CACHE                       */
} nnb_config;

typedef struct nnb_search_params {
    int32_t mode;        /* nnb_mode                                               */
    int32_t evaluator;   /* nnb_evaluator                                          */
    int32_t sims;        /* simulations per move (agent.num_simulations)           */
    int32_t weight_set;  /* resident weight set of the searcher (of colour 1 in an arena)  */
    int32_t flags;       /* NNB_FLAG_* bits                                        */
    int32_t rng_move_index; /* nnb_search only: move-index coordinate of the RNG   */
    int32_t rng_game_base;  /* nnb_search only: root i draws as game rng_game_base+i */
    int32_t weight_set_colour2_plus1; /* 0: searches for colour 2 use weight_set too; k > 0: they use
                                         weight set k-1 (two-network arena, run_tournament.py:12-42) */
    double  c_puct;      /* exploration constant (agent.c_puct, default 1.4)       */
    double  dirichlet_alpha;   /* AZ only; 0 disables root noise                   */
    double  dirichlet_eps;     /* AZ only; mixing weight of the noise              */
    uint64_t seed;       /* AZ noise / ROLLOUT playouts: counter-based RNG key     */
    /* COMPAT, nnb_selfplay_round only: the colour whose stones fill the network's first input
     * plane (agent.current_player, agent_code.py:85-87) while colour 1 / colour 2 is to move;
     * 0 = the mover, which is what self_play keeps (train_alpha_zero.py:38,45).  {1, 1} is an
     * agent whose current_player is never updated, as in run_tournament.py:21-22. */
    int32_t side_of_colour[2];
    /* nnb_selfplay_round only: give up with NNB_ERR_STATE once this many lock-step rounds (tree step +
     * forward pass) have run and games are still unfinished; 0 = only the built-in bound, which no
     * legal round reaches.  Unfinished games keep result = 2 in their records. */
    int64_t round_limit;
} nnb_search_params;

/* One finished (or cut-off) self-play game.  Fixed size so records can be
 * all-gathered across ranks as raw bytes.  What train_alpha_zero.py:33-55 and
 * agent_code.py:254-267 keep per game is derivable from it. */
typedef struct nnb_ply {
    uint64_t colour_boards[2];   /* position BEFORE the move                        */
    uint16_t visits[7];          /* root child visit counts (agent_code.py:218-229) */
    uint8_t  action;             /* column played                                   */
    uint8_t  player;             /* colour that moved (1 or 2)                      */
} nnb_ply;                       /* 32 bytes */

typedef struct nnb_trajectory {
    int32_t  length;             /* plies played                                    */
    int32_t  result;             /* get_result(): +1 colour 1 won, -1 colour 2, 0 draw; 2 = unfinished */
    int32_t  game_id;            /* global game index                               */
    int32_t  reserved;
    uint64_t final_boards[2];    /* colour boards after the last move               */
    nnb_ply  plies[42];
} nnb_trajectory;                /* 32 + 42*32 = 1376 bytes */

typedef struct nnb_round_stats {
    int64_t games;               /* games played                                    */
    int64_t plies;               /* total moves over all games                      */
    int64_t sims;                /* total simulations                               */
    int64_t net_leaves;          /* leaves sent through the network                 */
    int64_t terminal_hits;       /* simulations that ended on a terminal node       */
    int64_t depth_sum;           /* sum of selection depths over all simulations    */
    int64_t steps;               /* lock-step iterations (tree kernel + forward)    */
    int64_t moves_rounds;        /* move rounds (<= 42)                             */
    int64_t cache_hits;          /* simulations answered by the evaluation cache    */
    int64_t chains;              /* tree step -> forward pass chains a lock-step round was split in (1, or 2
                                    halves on two streams): kernel launches per class = steps * chains      */
    int64_t finished_total;      /* nnb_selfplay_stream: games finished since the stream was (re)started    */
} nnb_round_stats;

/* ---- library ----------------------------------------------------------- */
const char *nnb_version(void);
const char *nnb_last_error(void);

/* ---- engine life cycle -------------------------------------------------- */
/* Bytes of device workspace an engine of this configuration needs. */
int nnb_workspace_bytes(const nnb_config *cfg, size_t *bytes_out);

/* Builds an engine inside `workspace` (device memory, 256-byte aligned, at least
 * nnb_workspace_bytes() long, zero-filled by this call).  Synchronous.
 * Replaces: AlphaZeroAgent.__init__ (agent_code.py:27-59) as far as device state goes. */
int nnb_engine_create(const nnb_config *cfg, void *workspace, size_t workspace_bytes,
                      void *stream, nnb_engine **engine_out);
int nnb_engine_destroy(nnb_engine *engine);

/* Repacks the ten float32 state-dict tensors of Connect4Net (network.py:28-41, keys
 * conv_layers.{0,2,4}.{weight,bias}, fc_layers.0.{weight,bias},
 * policy_head.{weight,bias}, value_head.{weight,bias}; device pointers, PyTorch
 * layouts) into the engine's kernel layouts.  Replaces: load_agent_model's
 * load_state_dict (utils.py:40-46). */
int nnb_load_weights(nnb_engine *engine, int weight_set, void *stream,
                     const float *conv0_w, const float *conv0_b,
                     const float *conv2_w, const float *conv2_b,
                     const float *conv4_w, const float *conv4_b,
                     const float *fc_w, const float *fc_b,
                     const float *policy_w, const float *policy_b,
                     const float *value_w, const float *value_b);

/* ---- K1: game rules on bitboards ---------------------------------------- */
/* For each of n positions: if actions[i] >= 0, drop a stone of colour players[i]
 * in that column (ConnectFourGame.make_move, connect_four_game.py:38-52: a full
 * column changes nothing).  Then report, for the resulting position,
 *   legal_out[i]  : 7-bit mask of open columns (get_valid_locations, :109-125)
 *   status_out[i] : bit0 = move applied, bit1 = colour 1 has four, bit2 = colour 2
 *                   has four (check_win, :148-192), bit3 = board full (:194-200);
 *                   is_terminal (:72-78) = any of bits 1-3.
 * actions may be NULL (no move; bit0 is then 0). */
int nnb_game_ops(void *stream, int32_t n, const uint64_t *colour_boards_in, const int8_t *players,
                 const int8_t *actions, uint64_t *colour_boards_out, uint8_t *legal_out,
                 uint8_t *status_out);

/* K3: side boards -> float32 planes [n,2,6,7] (AlphaZeroAgent.preprocess,
 * agent_code.py:76-89; row 0 of a plane is the top row). */
int nnb_encode_planes(void *stream, int32_t n, const uint64_t *side_boards, float *planes_out);

/* Selection score of mcts.py:29 in float64, element-wise:
 * value/(visits+1) + c_puct*sqrt(log(parent_visits+1)/(visits+1)), with log taken
 * from the engine's table (filled by the host's libm) for parent_visits <= max_sims. */
int nnb_ucb_scores(nnb_engine *engine, void *stream, int32_t n, const double *value_sum,
                   const int32_t *visits, const int32_t *parent_visits, double c_puct,
                   double *scores_out);

/* ---- K4-K6: policy-value network ---------------------------------------- */
/* Connect4Net.forward (network.py:43-50) for n <= max_games positions given as
 * side boards.  log_policy_out [n,7] and value_out [n] as the reference returns
 * them.  If legal_masks != NULL, priors_out [n,7] receives exp(log_policy) with
 * illegal columns zeroed and renormalised, uniform over legal columns if the sum
 * is 0 (agent_code.py:199-208).  Any output pointer may be NULL. */
int nnb_net_forward(nnb_engine *engine, int weight_set, void *stream, int32_t n,
                    const uint64_t *side_boards, const uint8_t *legal_masks,
                    float *log_policy_out, float *value_out, float *priors_out);

/* Same from float32 planes [n,2,6,7] whose entries are exactly 0 or 1. */
int nnb_net_forward_planes(nnb_engine *engine, int weight_set, void *stream, int32_t n,
                           const float *planes, float *log_policy_out, float *value_out);

/* Test hook: the hidden activations the last nnb_net_forward / nnb_net_forward_planes call left in the
 * engine for its first n positions, converted to float32 whatever the engine's arithmetic: layer 2 =
 * ReLU(conv_layers.2) as [n][8][9][128] (row, column, channel), layer 3 = ReLU(conv_layers.4) as
 * [n][9][10][128], layer 4 = ReLU(fc_layers.0) as [n][1024] (network.py:31-39,45-47).
 * No counterpart in the reference (torch keeps these inside forward). */
int nnb_debug_activations(nnb_engine *engine, int weight_set, void *stream, int layer, int32_t n, float *out);

/* ---- baseline opponent ---------------------------------------------------- */
/* MinimaxAgent.select_move (nnbattle/agents/minimax/agent_code.py:23-89, position score of
 * connect_four_game.py:215-278) for n positions at once: depth-limited minimax, depth in [0, 8],
 * for the side to move movers[i].  The reference agent maximises for colour 2 whoever moves; this
 * one maximises for the mover, which is the same search whenever colour 2 is to move.
 * column_out[i] = chosen column (lowest column among equally good ones; -1 in a finished position,
 * without a legal move, or at depth 0); where the reference falls back to random.choice because
 * every move loses inside the horizon, the lowest legal column is returned.  score_out[i] = root
 * value, +-2^30 for a forced win / loss.  Either output may be NULL.  Asynchronous on `stream`. */
int nnb_minimax(void *stream, int32_t n, const uint64_t *colour_boards, const int8_t *movers, int32_t depth,
                int8_t *column_out, int32_t *score_out);

/* ---- K2 + K7: search ----------------------------------------------------- */
/* Runs one search per root position (AlphaZeroAgent.mcts_simulate,
 * agent_code.py:144-234) for n <= max_games independent roots, one wavefront per
 * root.  visits_out [n,7] = root child visit counts (0 where no child),
 * action_out [n] = most visited child, lowest column on ties, -1 if the root got no
 * children (the reference returns None, agent_code.py:219-221).
 * root_value_out [n] (may be NULL) = the root's accumulated value sum.
 * root_players [n] = the colour to move at each root (env.current_player, agent_code.py:167):
 * the colour COMPAT places at every step of the descent.  side_players [n] (may be NULL =
 * root_players) = the colour the leaf planes are encoded from (agent.current_player,
 * agent_code.py:85-87), which the reference reads independently of the mover.  AZ mode always
 * encodes a leaf from the side to move there and ignores side_players.
 * root_noise [n,7] (AZ mode, may be NULL): noise mixed into the root priors with weight
 * dirichlet_eps; when NULL and dirichlet_alpha > 0 the engine draws Dirichlet(alpha)
 * noise itself from its counter-based RNG (see nnb_dirichlet_noise).
 * stats_out (host pointer, may be NULL) is filled when the call returns.  n = 0 is a no-op.  Synchronous. */
int nnb_search(nnb_engine *engine, void *stream, const nnb_search_params *params, int32_t n,
               const uint64_t *root_colour_boards, const int8_t *root_players,
               const int8_t *side_players, const float *root_noise, int32_t *visits_out, int8_t *action_out,
               double *root_value_out, nnb_round_stats *stats_out);

/* nnb_search for callers that cannot allocate device memory - code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The
 * search state that nnb_search keeps in the engine's workspace lives in a fixed array inside the
 * library instead (512 KB, one copy per GPU, held until the process ends), so there is no engine, no
 * workspace and no call to hipMalloc / hipFree or to any allocating routine; the first call on a
 * device lays that array out and fills its log table, WAITS for that set-up on `stream` (so later
 * calls on any stream find it done) and is refused with NNB_ERR_STATE while `stream` is capturing:
 * make one eager call per device before capturing.  Limits: n <= NNB_STATIC_MAX_ROOTS roots,
 * params->sims <= NNB_STATIC_MAX_SIMS, a fixed evaluator (CONST, HASH_Q8, HASH_F32 or ROLLOUT - the
 * network's activations do not fit), params->flags == 0; anything beyond is refused with NNB_ERR_ARG.
 * Arguments and results are those of nnb_search, bit for bit; there are no statistics, and the call is
 * ASYNCHRONOUS on `stream` on the current device (the first call on a device excepted).  All calls on one GPU share that GPU's array: calls
 * that overlap in time on one GPU (two streams, two threads) race and must be serialised by the caller. */
#define NNB_STATIC_MAX_ROOTS 256
#define NNB_STATIC_MAX_SIMS 800
int nnb_search_static(void *stream, const nnb_search_params *params, int32_t n,
                      const uint64_t *root_colour_boards, const int8_t *root_players,
                      const int8_t *side_players, const float *root_noise, int32_t *visits_out,
                      int8_t *action_out, double *root_value_out);

/* Plays n_games self-play games to the end, the same searcher on both sides (self_play,
 * train_alpha_zero.py:33-55), or one weight set per colour (run_tournament.py:12-42).
 * At most max_games games are in flight, all advancing in lock step; when n_games is
 * larger, a game that ends hands its slot to the next unplayed game, so the leaf batch
 * stays full until the round runs out of games.  Game i starts from the empty board, or
 * from start_colour_boards[i]/start_players[i] when given, and is recorded in
 * trajectories_out[i] (device memory, n_games records, plies past `length` zeroed) with
 * game_id = first_game_id + i;
 * what a game plays depends on its id and start position only, not on the slot or the
 * moment it was seated.  stats_out is a host pointer.  Synchronous. */
int nnb_selfplay_round(nnb_engine *engine, void *stream, const nnb_search_params *params,
                       int32_t n_games, int32_t first_game_id,
                       const uint64_t *start_colour_boards, const int8_t *start_players,
                       nnb_trajectory *trajectories_out, nnb_round_stats *stats_out);

/* Self-play as an endless stream of games: what the training loop's repeated self_play calls
 * (train_alpha_zero.py:33-55, called once per iteration from :57-119) amount to when no iteration waits
 * for the longest game of the one before.  All max_games slots stay occupied: game j = 0, 1, 2, ... has
 * game_id = first_game_id + j, starts from start position j % n_starts (the empty board when
 * start_colour_boards is NULL) and is recorded in record_ring[j % ring_len]; a slot whose game ends takes
 * the next j.  What a game plays depends on its id and start position only, so record j equals what
 * nnb_selfplay_round writes for the same id and start position, byte for byte.
 * restart != 0 seats games 0 .. max_games-1 (and keeps the pointer arguments, which must stay valid,
 * for the calls that follow); restart == 0 carries on with the games seated by the previous call
 * (params may be NULL or must equal the first call's; the pointer arguments are ignored).  Either way
 * the call returns once at least min_finished more games have finished; games in flight stay seated.
 * finished_ring (device, ring_len ints, may be NULL): entry k % ring_len = index into record_ring of the
 * k-th game to finish since the restart; with stats_out->finished_total = F after the call and
 * stats_out->games = f, this call's records are entries F-f .. F-1.  ring_len >= 4 * max_games and
 * min_finished <= ring_len - 2 * max_games (a call's records must fit the ring beside the games in flight); a record is
 * reused ring_len games later, so the caller copies finished records out between calls (a game that lasts
 * while ring_len - max_games later games are seated would be overwritten; no legal game does at 4 x).
 * The host looks at the device's counters every 64 lock-step rounds only, so a call can overshoot min_finished;
 * if more than ring_len - max_games games finish inside one call (searches of a handful of simulations, or
 * start positions that are mostly finished) records it was about to report have been overwritten: the call
 * then fails with NNB_ERR_STATE and the stream is ended.
 * stats_out counts the work of this call only.  Any other call that searches or runs the network on this
 * engine ends the stream (the next call needs restart != 0).  nnb_load_weights between two calls does not:
 * the games in flight carry on and every later leaf is judged by the new weights, as when the training loop
 * updates the model between two self_play calls (entries a game has filed in the evaluation cache under the
 * old weights stay until that game ends, so leave NNB_FLAG_EVAL_CACHE off when weights change).  The per-slot
 * statistics behind stats_out are 32-bit: restart the stream before a slot has run 2^31 simulations (about
 * 9 * 10^5 games per slot at 200 simulations per move).  Synchronous. */
int nnb_selfplay_stream(nnb_engine *engine, void *stream, const nnb_search_params *params, int32_t restart,
                        int32_t first_game_id, int32_t min_finished, const uint64_t *start_colour_boards,
                        const int8_t *start_players, int32_t n_starts, nnb_trajectory *record_ring,
                        int32_t *finished_ring, int32_t ring_len, nnb_round_stats *stats_out);

/* Records -> per-ply training samples on the device: the per-move targets AlphaZeroAgent.self_play INTENDS
 * (agent_code.py:254-267: state, MCTS visit distribution, value target) in the form ConnectFourDataset
 * stacks (data_module.py:13-27).  Not literal parity with that (dead) method, and no reference fixture covers
 * it: the reference stores the raw board (state.copy(), agent_code.py:256), not planes, and flips its player
 * variable with `player = -player` (agent_code.py:266), so the second mover's target is -1 in every decided
 * game.  What is built here is the mover-relative target below; the only pin is this repository's own host
 * loop (alphazero_amd.selfplay.self_play_samples), which the kernel equals bit for bit.
 * For ply k < length of record i, sample s = ply_offsets[i] + k:
 *   planes_out[s]  float32 [2,6,7]: plane 0 the mover's stones, plane 1 the rival's, row 0 = top
 *                  (preprocess, agent_code.py:76-89), the position BEFORE the move;
 *   probs_out[s]   float32 [7]: root visit counts / their sum (agent_code.py:226-232; zeros if no visits);
 *   z_out[s]       float32: +1 if the mover of that ply won the game, -1 if it lost, 0 for a draw or an
 *                  unfinished record (agent_code.py:259-266).
 * ply_offsets [n] (device) = number of plies in the records before record i (exclusive prefix sum of the
 * lengths, which the caller computes); the outputs hold sum(length) samples.  Asynchronous on `stream`. */
int nnb_trajectory_samples(void *stream, int32_t n, const nnb_trajectory *records, const int32_t *ply_offsets,
                           float *planes_out, float *probs_out, float *z_out);

/* The Dirichlet(alpha) root noise the engine draws for (seed, game id, move index) over the
 * legal columns of each row: out [n,7], zero on illegal columns, rows sum to 1.  Exposed so a
 * caller (or a test) can reproduce exactly what an AZ-mode search mixed into its root.
 * No counterpart in the reference (its search has no noise). */
int nnb_dirichlet_noise(void *stream, int32_t n, uint64_t seed, const int32_t *game_ids,
                        const int32_t *move_indices, const uint8_t *legal_masks, double alpha,
                        float *noise_out);

/* Average device time, in milliseconds, of the `which`-th engine kernel class since
 * the last nnb_profile_reset, measured with HIP events on the launch stream when
 * profiling is enabled (nnb_profile_enable(engine, 1)).  Classes: 0 tree step,
 * 1 conv0, 2 conv2, 3 conv4, 4 fc, 5 heads, 6 move readout. */
int nnb_profile_enable(nnb_engine *engine, int on);
int nnb_profile_reset(nnb_engine *engine);
int nnb_profile_get(nnb_engine *engine, int which, double *avg_ms_out, int64_t *launches_out);
/* Shader clock, in MHz, that the last profiled launch of the convolution kernel (mhz_out[0]) and of the linear
 * layer (mhz_out[1]) ran at: block 0 of those kernels stamps the shader-cycle counter and the constant 100 MHz
 * clock at its start and end while profiling is enabled (float32-faithful path; 0 when nothing was stamped).
 * The matrix peak a roofline is quoted against assumes 2400 MHz.  Synchronous.  No reference counterpart. */
int nnb_profile_clocks(nnb_engine *engine, void *stream, double *mhz_out);
/* The engine runs the two chains of a self-play round (and the two networks of an arena) on the caller's stream and
 * on a second stream of its own.  HIP spreads a process's streams over a few hardware queues, and two streams on one
 * queue run their kernels one after the other; which queue a stream gets depends on the streams the process already
 * has (PyTorch's, RCCL's).  Before its first launch on a pair, every such call therefore measures whether the pair
 * runs side by side (two 200 us spin kernels) and replaces the second stream until it does.  This entry point runs
 * that check for `stream` now and measures again: returns 1 if the pair runs side by side, 0 if not, negative on
 * error; *replaced_out (may be null) = second streams discarded since the engine was created.  Synchronous.  No
 * reference counterpart. */
int nnb_debug_second_stream(nnb_engine *engine, void *stream, int *replaced_out);

#ifdef __cplusplus
}
#endif
#endif /* NNB_H */

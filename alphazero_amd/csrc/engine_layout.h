// Device-side layout of an engine's workspace, shared by the kernels and the host API.
#pragma once
#include <stdint.h>
#include "../../include/nnb.h"

namespace nnb {

typedef unsigned long long u64;

// One MCTS node, 16 bytes so a child is one 128-bit load.  Children of a node are
// allocated contiguously in ascending column order (mcts.py:35-38 iterates
// legal_moves, which connect_four_game.py:115 builds ascending), so a node only
// stores the index of its first child and how many it has.
struct __attribute__((aligned(16))) Node {
    double   value;        // accumulated backup value (MCTSNode.value, mcts.py:20)
    int32_t  visits;       // MCTSNode.visits (mcts.py:19)
    uint16_t first_child;  // index of the first child in this game's pool
    uint8_t  nchild;       // 0 = leaf (MCTSNode.is_leaf, mcts.py:22-23)
    uint8_t  action;       // column that leads here (MCTSNode.action)
};
static_assert(sizeof(Node) == 16, "Node must be 16 bytes");

constexpr int kPathSlots = 64;      // one path entry per lane; depth <= 42
constexpr int kMaxPlies = 42;

// Pointers into the workspace (all device memory).  G = max_games.
struct CacheEntry {             // 64 bytes
    u64 me, opp;                // side boards as the evaluator saw them
    float value;
    uint32_t gen;
    float prior[7];             // masked, renormalised priors (before any root noise)
    uint32_t pad[3];
};

struct EngineDev {
    int32_t  max_games;
    int32_t  pool_cap;          // nodes per game = 1 + 7*max_sims
    int32_t  max_sims;
    int32_t  n_games;           // games / roots in the current call

    Node    *nodes;             // [G][pool_cap]
    float   *prior;             // [G][pool_cap]   (AZ mode only)
    double  *log_table;         // [max_sims + 8]  log(n + 1), filled by the host's libm

    u64     *root_boards;       // [G][2] colour boards of each root
    int8_t  *root_player;       // [G] colour to move at the root (env.current_player, agent_code.py:167)
    int8_t  *root_side;         // [G] colour the leaf planes are encoded from (agent.current_player,
                                //     agent_code.py:85-87); 0 = SearchDev::side of the mover, else the mover
    int32_t *active;            // [G] 1 while the game / search is live
    int32_t *n_nodes;           // [G]
    int32_t *sims_done;         // [G]
    int32_t *pending;           // [G] leaf-batch slot awaiting its value, -1 = none
    int32_t *path_len;          // [G]
    uint16_t *path;             // [G][kPathSlots] node indices root -> leaf of the pending leaf
    int32_t *ply;               // [G] plies played (self-play)
    int32_t *game_stats;        // [G][8]: depth_sum, terminal_hits, net_leaves, sims, plies of retired games
    int32_t *traj_index;        // [G] which trajectory record (game of the round) the slot is playing
    int32_t *game_id;           // [G] global game index (RNG coordinate)
    int32_t *move_index;        // [G] move index of the current search (RNG coordinate)
    float   *root_noise;        // [G][7] Dirichlet noise mixed into the root priors (AZ)

    // Compacted leaf batches.  There are two of them (batch b occupies slots [b*G, (b+1)*G)):
    // when the two colours are played by different weight sets, leaves of searches whose
    // root mover is colour 1 go to batch 0 and those of colour 2 to batch 1; otherwise only
    // batch 0 is used.
    u64     *leaf_side;         // [2G][2] side boards (me, opp)
    uint8_t *leaf_legal;        // [2][2G] legal mask of each batched leaf, indexed by [step parity][slot]: a step
                                //         reads the masks of the leaves it consumes while other blocks already
                                //         write those of the next batch
    int32_t *leaf_count;        // [2][2]  batch sizes, indexed by [step parity][batch]
    float   *leaf_value;        // [2G]    network / evaluator value per batch slot
    float   *leaf_logp;         // [2G][7] network log-policy per batch slot
    float   *leaf_prior;        // [2G][7] masked, renormalised priors per batch slot
    // Evaluation cache (optional): per game a direct-mapped table of the leaves it has evaluated.
    // An entry is valid when its generation equals cache_gen[g], which seating a new game bumps.
    CacheEntry *cache;          // [G][cache_slots] or null
    uint32_t *cache_gen;        // [G]
    u64     *pend_key;          // [G][2] side boards of the game's pending leaf (to file its evaluation)
    int32_t cache_slots;
    int32_t *counters;          // [16]   [0] = live games after the last readout
    unsigned long long *stats64; // [8]   depth, terminal hits, leaves, sims, plies (stats_reduce_kernel)
};

// A self-play round: `total_games` games are played on n_games <= total_games slots; a slot whose
// game ends takes the next unplayed game, n_games + counters[kRefills]++, until none is left.
// Game j starts from start position j % n_starts and is recorded in traj[j % ring_len] (a plain round
// has n_starts = ring_len = total_games; an endless stream of games wraps around both).
struct RoundDev {
    const u64 *start_boards;    // [n_starts][2] or null (empty boards)
    const int8_t *start_players; // [n_starts] or null (colour 1 starts)
    nnb_trajectory *traj;       // [ring_len]
    int32_t total_games;
    int32_t first_game_id;
    int32_t ring_len;
    int32_t n_starts;
    int32_t *finished_ring;     // [ring_len] or null: entry k % ring_len = record index of the k-th game to finish
};
constexpr int kRefills = 16;    // index into EngineDev::counters; entries 0..15 are cleared every move
constexpr int kLive = 17;       // slots that still hold a game (kept through a whole round)
constexpr int kEndlessGames = 1 << 30;   // total_games at or above this: an endless stream of games (nnb_selfplay_stream)
constexpr int kSeatTries = 4096;         // finished start positions in a row after which a slot of a stream gives up
constexpr int kSyncErr = 19;    // waits between blocks of the split linear layer that ran out (results invalid if != 0)
constexpr int kFinished = 18;   // games finished since the round was set up (directly behind kLive: read together)

// Games [first, end) and the leaf batch they file their leaves in (-1: by the root mover's weight set).
struct GameSpan {
    int32_t first, end, batch;
};

struct SearchDev {
    int32_t mode;
    int32_t evaluator;
    int32_t sims;
    int32_t flags;
    int32_t rng_move_index;
    int32_t rng_game_base;
    int32_t wset[2];            // weight set searching for colour 1 / colour 2
    int32_t side[2];            // COMPAT: colour the planes are encoded from while colour 1 / 2 moves; 0 = the mover
    double  c_puct;
    double  dirichlet_alpha;
    double  dirichlet_eps;
    u64     seed;
};

}  // namespace nnb

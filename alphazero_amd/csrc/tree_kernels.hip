// Game-rule, search-tree and move-readout kernels (gfx950, wave64).
//
// One wavefront owns one game's search tree.  Lanes 0..6 score the (at most
// seven) children of the current node in float64, an 8-lane butterfly picks the
// best with the lowest column winning ties, and the path from the root is kept
// one node per lane so the backup is a single atomic-free read-modify-write per
// lane.  Compile with -ffp-contract=off: the selection score must round exactly
// like the CPython expression it replaces (nnbattle/agents/alphazero/mcts.py:29).
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include "c4_bits.h"
#include "engine_layout.h"
#include "kernels.h"

namespace nnb {

using c4::u64;

// ---------------------------------------------------------------------------
// K1: rules on bitboards.  Replaces ConnectFourGame.make_move / get_valid_locations
// / check_win / is_board_full (connect_four_game.py:38-52,109-125,148-200).
// ---------------------------------------------------------------------------
__global__ void game_ops_kernel(int n, const u64 *__restrict__ in, const int8_t *__restrict__ players,
                                const int8_t *__restrict__ actions, u64 *__restrict__ out,
                                uint8_t *__restrict__ legal_out, uint8_t *__restrict__ status_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 b1 = in[2 * i], b2 = in[2 * i + 1];
    unsigned status = 0;
    if (actions) {
        int a = actions[i];
        if (a >= 0 && a < 7 && ((c4::legal_mask(b1 | b2) >> a) & 1)) {
            u64 cell = c4::drop_cell(b1 | b2, a);
            if (players[i] == 1) b1 |= cell; else b2 |= cell;
            status |= 1;
        }
    }
    if (c4::has_four(b1)) status |= 2;
    if (c4::has_four(b2)) status |= 4;
    if (c4::is_full(b1 | b2)) status |= 8;
    if (out) { out[2 * i] = b1; out[2 * i + 1] = b2; }
    if (legal_out) legal_out[i] = (uint8_t)c4::legal_mask(b1 | b2);
    if (status_out) status_out[i] = (uint8_t)status;
}

// K3: side boards -> float planes [n,2,6,7], row 0 = top (agent_code.py:84-88).
__global__ void encode_planes_kernel(int n, const u64 *__restrict__ side, float *__restrict__ planes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per output element
    if (i >= n * 84) return;
    int b = i / 84, r = i % 84;
    int pl = r / 42, cell = r % 42, row = cell / 7, col = cell % 7;
    u64 bits = side[2 * b + pl];
    planes[i] = (float)((bits >> (col * 7 + (5 - row))) & 1);
}

// Finished records -> per-ply training samples, one thread per (record, ply): planes from the mover's side
// (agent_code.py:84-88), the root visit distribution (visits / their sum, the division in float64 as numpy does
// it, agent_code.py:226-232), and z = +1 / -1 / 0 as the mover of that ply went on to win / lose / draw
// (agent_code.py:259-266).  ply_offsets[i] = plies of the records before record i.
__global__ void trajectory_samples_kernel(int n, const nnb_trajectory *__restrict__ rec,
                                          const int32_t *__restrict__ ply_offsets, float *__restrict__ planes,
                                          float *__restrict__ probs, float *__restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kMaxPlies) return;
    const int g = i / kMaxPlies, k = i - g * kMaxPlies;
    const nnb_trajectory *t = rec + g;
    int len = t->length;
    len = len < 0 ? 0 : (len > kMaxPlies ? kMaxPlies : len);
    if (k >= len) return;
    const nnb_ply *ply = &t->plies[k];
    const size_t o = (size_t)ply_offsets[g] + k;
    const int me = ply->player == 2 ? 2 : 1;
    const u64 mine = ply->colour_boards[me - 1], theirs = ply->colour_boards[2 - me];
    float *pl = planes + o * 84;
    for (int cell = 0; cell < 42; ++cell) {
        const int row = cell / 7, col = cell - row * 7, bit = col * 7 + (5 - row);
        pl[cell] = (float)((mine >> bit) & 1);
        pl[42 + cell] = (float)((theirs >> bit) & 1);
    }
    double sum = 0.0;
    for (int c = 0; c < 7; ++c) sum += (double)ply->visits[c];
    for (int c = 0; c < 7; ++c) probs[o * 7 + c] = sum > 0.0 ? (float)((double)ply->visits[c] / sum) : 0.f;
    const int res = t->result;
    z[o] = (res == 1 || res == -1) ? (((res == 1) == (me == 1)) ? 1.f : -1.f) : 0.f;
}

// mcts.py:29 element-wise, float64.
__global__ void ucb_scores_kernel(int n, const double *__restrict__ value, const int32_t *__restrict__ visits,
                                  const int32_t *__restrict__ parent_visits, double c_puct,
                                  const double *__restrict__ log_table, int log_n, double *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = (double)(visits[i] + 1);
    int pv = parent_visits[i];
    double lg = pv < log_n ? log_table[pv] : log((double)(pv + 1));
    out[i] = (value[i] / d) + c_puct * sqrt(lg / d);
}

// ---------------------------------------------------------------------------
// Wave-level search primitives.  A pool is anything indexed by node number: a Node* (global memory or LDS; after
// inlining the compiler resolves the address space) or a SplitArray, whose first nodes live in LDS and the rest in
// memory.
// ---------------------------------------------------------------------------
// Elements 0 .. cap-1 in LDS, the others in memory at their own index: the top of a search tree is created first and
// visited most, so a tree too large for its share of the LDS keeps its hot part there.
template <typename T>
struct SplitArray {
    T *lds;
    T *mem;
    int cap;
    __device__ __forceinline__ T &operator[](int i) const { return *(i < cap ? lds + i : mem + i); }
};

struct Descent {
    u64 c1, c2;     // colour boards at the leaf
    int node;       // leaf node index
    int len;        // nodes on the path, root included
    bool terminal;  // leaf position is terminal (connect_four_game.py:72-78)
};

// Selection, agent_code.py:170-176 with MCTSNode.best_child (mcts.py:25-33).
// `player` is the root mover: the reference flips the mover in make_move
// (connect_four_game.py:50) and flips it back at agent_code.py:176, so every stone
// placed during descent has the root mover's colour.  `mypath` is this lane's entry of
// the path (lane i holds the i-th node).
template <typename Pool>
__device__ __forceinline__ Descent descend_compat(const Pool &pool, const double *logt, double c_puct,
                                                  int player, u64 c1, u64 c2, int lane, int &mypath) {
    Descent r;
    Node root = pool[0];
    int cur_visits = root.visits, first = root.first_child, nch = root.nchild;
    r.node = 0; r.len = 1; mypath = 0;
    r.terminal = c4::is_terminal(c1, c2);
    while (nch > 0 && !r.terminal) {
        const bool in = lane < nch;
        Node ch = pool[first + (in ? lane : 0)];
        double d = (double)(ch.visits + 1);
        double score = (ch.value / d) + c_puct * sqrt(logt[cur_visits] / d);
        if (!in) score = -INFINITY;
        int best = lane;
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            double os = __shfl_xor(score, off);
            int ob = __shfl_xor(best, off);
            bool take = (os > score) || (os == score && ob < best);   // strict '>' at mcts.py:30
            score = take ? os : score;
            best = take ? ob : best;
        }
        best = __shfl(best, 0);
        int a = __shfl((int)ch.action, best);
        cur_visits = __shfl(ch.visits, best);
        int nfirst = __shfl((int)ch.first_child, best);
        int nn = __shfl((int)ch.nchild, best);
        r.node = first + best;
        first = nfirst; nch = nn;
        u64 cell = c4::drop_cell(c1 | c2, a);
        if (player == 1) c1 |= cell; else c2 |= cell;
        if (lane == r.len) mypath = r.node;
        ++r.len;
        r.terminal = c4::is_terminal(c1, c2);
    }
    r.c1 = c1; r.c2 = c2;
    return r;
}

// MCTSNode.expand (mcts.py:35-38): one child per legal column, ascending; the
// reference discards the priors here.  Returns the number of children created.
template <typename Pool>
__device__ __forceinline__ int expand_compat(const Pool &pool, int node, int n_nodes, unsigned legal, int lane) {
    int nlegal = __popc(legal);
    if (lane < nlegal) {
        unsigned m = legal;
        for (int k = 0; k < lane; ++k) m &= m - 1;        // drop the `lane` lowest set bits
        Node c;
        c.value = 0.0; c.visits = 0; c.first_child = 0; c.nchild = 0;
        c.action = (uint8_t)(__ffs(m) - 1);
        pool[n_nodes + lane] = c;
    }
    if (lane == 0) {
        pool[node].first_child = (uint16_t)n_nodes;
        pool[node].nchild = (uint8_t)nlegal;
    }
    return nlegal;
}

// MCTSNode.backpropagate (mcts.py:40-44): +1 visit and the same un-negated value on
// every node of the path; one node per lane, so no atomics.
template <typename Pool>
__device__ __forceinline__ void backup_path(const Pool &pool, int mypath, int len, double v, int lane) {
    if (lane < len) {
        Node &n = pool[mypath];
        n.visits += 1;
        n.value += v;
    }
}

__device__ __forceinline__ void wave_mem_sync() {
    // lanes of this wave read nodes other lanes of the same wave just wrote
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The fixed evaluators of weight set k > 0 see the position through a different hash, so a
// two-set arena can be checked bit for bit without networks.
__device__ __forceinline__ u64 set_mix(int wset) { return (u64)wset * 0xD6E8FEB86659FD93ULL; }

// The colour whose stones fill the first input plane of a COMPAT leaf (agent.current_player,
// agent_code.py:85-87): the caller's per-root choice, else the round's per-colour choice, else the mover.
__device__ __forceinline__ int view_side(const EngineDev &e, const SearchDev &p, int g, int player) {
    int s = e.root_side[g];
    if (s == 0) s = p.side[player - 1];
    return s == 0 ? player : s;
}

__device__ __forceinline__ double fixed_eval(int evaluator, u64 me, u64 opp) {
    if (evaluator == NNB_EVAL_HASH_Q8) return c4::hash_value_q8(me, opp);
    if (evaluator == NNB_EVAL_HASH_F32) return c4::hash_value_f32(me, opp);
    return 0.0;
}

// ---------------------------------------------------------------------------
// Counter-based RNG (Philox4x32-10): integer only, so the CPU restatement reproduces
// every draw.  Counter = (game id, move index, stream, block), key = seed.
// ---------------------------------------------------------------------------
struct U32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// ROLLOUT evaluator: uniformly random playout to the end, one Philox word per ply
// (stream = simulation index).  Returns c4::result() of the final position.
__device__ __forceinline__ int rollout(u64 b1, u64 b2, int mover, u64 seed, uint32_t game, uint32_t move,
                                       uint32_t sim) {
    U32x4 rnd = {0, 0, 0, 0};
    for (int rp = 0; rp < 42; ++rp) {
        if (c4::is_terminal(b1, b2)) break;
        unsigned legal = c4::legal_mask(b1 | b2);
        if (!legal) break;
        if ((rp & 3) == 0) rnd = philox4x32(game, move, sim, (uint32_t)(rp >> 2), (uint32_t)seed, (uint32_t)(seed >> 32));
        uint32_t word = (rp & 3) == 0 ? rnd.x : ((rp & 3) == 1 ? rnd.y : ((rp & 3) == 2 ? rnd.z : rnd.w));
        int idx = (int)(((uint64_t)word * (uint64_t)__popc(legal)) >> 32);
        unsigned m = legal;
        for (int k = 0; k < idx; ++k) m &= m - 1;
        u64 cell = c4::drop_cell(b1 | b2, __ffs(m) - 1);
        if (mover == 1) b1 |= cell; else b2 |= cell;
        mover = 3 - mover;
    }
    return c4::result(b1, b2);
}

// Priors of the fixed evaluators for column `col`: uniform over the legal columns, or
// weights 1 + byte_col of the position hash, normalised in float32 in ascending order.
__device__ __forceinline__ float fixed_prior(int evaluator, u64 me, u64 opp, unsigned legal, int col) {
    if (evaluator == NNB_EVAL_HASH_Q8 || evaluator == NNB_EVAL_HASH_F32) {
        u64 h = c4::mix64(me, opp);
        float sum = 0.0f, mine = 0.0f;
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            float w = ((legal >> a) & 1) ? (float)(1 + (int)((h >> (8 * a)) & 0xFF)) : 0.0f;
            sum += w;
            if (a == col) mine = w;
        }
        return mine / sum;
    }
    return (col < 7 && ((legal >> col) & 1)) ? (float)(1.0 / (double)__popc(legal)) : 0.0f;
}

// Dirichlet(alpha) over the legal columns: Marsaglia-Tsang gamma variates from Philox
// uniforms (stream 0xD1D1D1D1), normalised.  out[c] = 0 for illegal columns.
struct NoiseRng {
    uint32_t game, move, block; u64 seed; U32x4 buf; int have;
    __device__ double uniform() {           // (0,1), 53 bits
        if (have == 0) { buf = philox4x32(game, move, 0xD1D1D1D1u, block++, (uint32_t)seed, (uint32_t)(seed >> 32)); have = 2; }
        uint32_t a = have == 2 ? buf.x : buf.z, b = have == 2 ? buf.y : buf.w;
        --have;
        return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) / 9007199254740992.0;
    }
    __device__ double normal() {
        double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    __device__ double gamma(double alpha) {
        double boost = 1.0;
        if (alpha < 1.0) { boost = pow(uniform(), 1.0 / alpha); alpha += 1.0; }
        const double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (int tries = 0; tries < 64; ++tries) {
            double x = normal(), v = 1.0 + c * x;
            if (v <= 0.0) continue;
            v = v * v * v;
            double u = uniform();
            if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
        }
        return d * boost;
    }
};

__device__ void dirichlet_noise(u64 seed, uint32_t game, uint32_t move, unsigned legal, double alpha, float *out) {
    NoiseRng rng{game, move, 0, seed, {0, 0, 0, 0}, 0};
    double gsum = 0.0, gv[7];
    for (int c = 0; c < 7; ++c) {
        gv[c] = ((legal >> c) & 1) ? rng.gamma(alpha) : 0.0;
        gsum += gv[c];
    }
    const int nl = __popc(legal);
    for (int c = 0; c < 7; ++c)
        out[c] = gsum > 0.0 ? (float)(gv[c] / gsum) : (((legal >> c) & 1) ? (float)(1.0 / nl) : 0.0f);
}

__global__ void dirichlet_noise_kernel(int n, u64 seed, const int32_t *__restrict__ game_ids,
                                       const int32_t *__restrict__ move_index, const uint8_t *__restrict__ legal,
                                       double alpha, float *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dirichlet_noise(seed, (uint32_t)game_ids[i], (uint32_t)move_index[i], legal[i], alpha, out + 7 * i);
}

// ---------------------------------------------------------------------------
// PUCT ("AZ") primitives.  W of a node is the value sum seen by the player who moved
// into it, so a parent maximises Q = W/N directly.
// ---------------------------------------------------------------------------
struct DescentAz {
    u64 c1, c2;
    int node, len, mover;   // mover = side to move at the leaf
    bool terminal;
};

template <typename Pool, typename Priors>
__device__ __forceinline__ DescentAz descend_az(const Pool &pool, const Priors &prior, double c_puct, int player,
                                                u64 c1, u64 c2, int lane, int &mypath) {
    DescentAz r;
    Node root = pool[0];
    int cur_visits = root.visits, first = root.first_child, nch = root.nchild;
    r.node = 0; r.len = 1; r.mover = player; mypath = 0;
    r.terminal = c4::is_terminal(c1, c2);
    while (nch > 0 && !r.terminal) {
        const bool in = lane < nch;
        const int ci = first + (in ? lane : 0);
        Node ch = pool[ci];
        const double n = (double)ch.visits;
        const double q = ch.visits > 0 ? ch.value / n : 0.0;
        const double u = c_puct * (double)prior[ci] * sqrt((double)cur_visits) / (1.0 + n);
        double score = q + u;
        if (!in) score = -INFINITY;
        int best = lane;
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            double os = __shfl_xor(score, off);
            int ob = __shfl_xor(best, off);
            bool take = (os > score) || (os == score && ob < best);
            score = take ? os : score;
            best = take ? ob : best;
        }
        best = __shfl(best, 0);
        int a = __shfl((int)ch.action, best);
        cur_visits = __shfl(ch.visits, best);
        int nfirst = __shfl((int)ch.first_child, best);
        int nn = __shfl((int)ch.nchild, best);
        r.node = first + best;
        first = nfirst; nch = nn;
        u64 cell = c4::drop_cell(c1 | c2, a);
        if (r.mover == 1) c1 |= cell; else c2 |= cell;
        r.mover = 3 - r.mover;
        if (lane == r.len) mypath = r.node;
        ++r.len;
        r.terminal = c4::is_terminal(c1, c2);
    }
    r.c1 = c1; r.c2 = c2;
    return r;
}

// Children with priors; `pr` is the prior of column `lane` (lanes 0..6).
template <typename Pool, typename Priors>
__device__ __forceinline__ int expand_az(const Pool &pool, const Priors &prior, int node, int n_nodes, unsigned legal, float pr,
                                         int lane) {
    const int nlegal = __popc(legal);
    unsigned m = legal;
    for (int k = 0; k < lane && k < 7; ++k) m &= m - 1;
    const int col = m ? __ffs(m) - 1 : 0;
    const float pc = __shfl(pr, col);
    if (lane < nlegal) {
        Node c;
        c.value = 0.0; c.visits = 0; c.first_child = 0; c.nchild = 0; c.action = (uint8_t)col;
        pool[n_nodes + lane] = c;
        prior[n_nodes + lane] = pc;
    }
    if (lane == 0) {
        pool[node].first_child = (uint16_t)n_nodes;
        pool[node].nchild = (uint8_t)nlegal;
    }
    return nlegal;
}

// v = value for the side to move at the leaf; the leaf gets -v, its parent +v, ...
template <typename Pool>
__device__ __forceinline__ void backup_az(const Pool &pool, int mypath, int len, double v, int lane) {
    if (lane < len) {
        Node &n = pool[mypath];
        n.visits += 1;
        n.value += ((len - 1 - lane) & 1) ? v : -v;
    }
}

// value of a leaf for the side to move, fixed evaluators (AZ frame)
__device__ __forceinline__ double fixed_eval_az(const SearchDev &p, u64 c1, u64 c2, int mover, uint32_t game,
                                                uint32_t move, uint32_t sim, u64 smix) {
    if (p.evaluator == NNB_EVAL_ROLLOUT) {
        int res = rollout(c1, c2, mover, p.seed, game, move, sim);
        return (res == 2 || res == 0) ? 0.0 : ((res == 1) == (mover == 1) ? 1.0 : -1.0);
    }
    return fixed_eval(p.evaluator, (mover == 1 ? c1 : c2) ^ smix, mover == 1 ? c2 : c1);
}

// ---------------------------------------------------------------------------
// Start of a move: every live game gets a fresh tree.  COMPAT: root.visits = 1
// (agent_code.py:159-160).  AZ: root.visits = 0 and, when alpha > 0, fresh Dirichlet
// noise for the root unless the caller supplied it.
// ---------------------------------------------------------------------------
__device__ void init_search(const EngineDev &e, const SearchDev &p, int g, int make_noise, int selfplay) {
    Node r;
    r.value = 0.0; r.visits = p.mode == NNB_MODE_AZ ? 0 : 1; r.first_child = 0; r.nchild = 0; r.action = 0xFF;
    e.nodes[(size_t)g * e.pool_cap] = r;
    e.n_nodes[g] = 1;
    e.sims_done[g] = 0;
    e.pending[g] = -1;
    e.path_len[g] = 0;
    e.move_index[g] = selfplay ? e.ply[g] : p.rng_move_index;
    if (make_noise) {
        unsigned legal = c4::legal_mask(e.root_boards[2 * g] | e.root_boards[2 * g + 1]);
        dirichlet_noise(p.seed, (uint32_t)e.game_id[g], (uint32_t)e.move_index[g], legal, p.dirichlet_alpha,
                        e.root_noise + 7 * g);
    }
}

__global__ void tree_init_kernel(EngineDev e, SearchDev p, int make_noise, int selfplay) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) { e.leaf_count[0] = 0; e.leaf_count[1] = 0; e.leaf_count[2] = 0; e.leaf_count[3] = 0; }
    if (g >= e.n_games) return;
    if (!e.active[g]) return;
    init_search(e, p, g, make_noise, selfplay);
}

// ---------------------------------------------------------------------------
// K2, lock-step form: one launch per simulation round.
//   1. consume the evaluator's output for this game's pending leaf: (AZ) expand the leaf
//      with its priors, then back the value up the stored path;
//   2. run simulations until one needs the evaluator (a non-terminal leaf) or the budget
//      is used up; terminal leaves are backed up on the spot (agent_code.py:182-186);
//   3. (COMPAT) expand the leaf - the reference ignores priors; append the leaf's side
//      boards to the compacted batch and remember the path for step 1 of the next launch.
// Launch with 1024 threads = 16 games per block, one wavefront per game.
// ---------------------------------------------------------------------------
constexpr int kStepGames = 16;                    // games (wavefronts) per block of the step kernel
// Simulations that end in a finished position need no network and are backed up on the spot,
// but at most this many per step: a game deep in a won position would otherwise run its whole
// budget inside one launch while the other 15 waves of its block (and the batch) wait.  The
// order of a game's simulations is unchanged, so results are too; S+1 steps still suffice.
#ifndef NNB_TERMINAL_SIMS_PER_STEP
#define NNB_TERMINAL_SIMS_PER_STEP 6
#endif
constexpr int kTerminalSimsPerStep = NNB_TERMINAL_SIMS_PER_STEP;

__device__ void readout_game(const EngineDev &e, const SearchDev &p, int g, int32_t *__restrict__ visits_out,
                             int8_t *__restrict__ action_out, double *__restrict__ root_value_out,
                             const RoundDev &round, int selfplay);

// HEADS: the network's heads for the pending leaf are computed here, by the wave that owns the
// game, from the linear layer's output - no separate heads launch, no round trip of values and
// priors through memory.  The head rows the block needs are staged in LDS first (the value row
// only under the reference rule, which ignores priors; all 8 rows under PUCT; per batch).
//
// TURNOVER (self-play without a common move boundary, no device-drawn root noise): a game whose
// simulation budget is used up ends its move here - readout, move, trajectory, retirement or the
// next game of the round, fresh root - before the step carries on with its next search.  With
// root noise to draw, turnover_kernel does the same ahead of the step instead.
//
// CACHE (NNB_FLAG_EVAL_CACHE): the evaluation of every leaf is filed in the game's direct-mapped
// table when it arrives; a leaf whose position is already there ends its simulation on the spot
// with the filed value and priors, exactly as the fresh evaluation would have a round later.
//
// SPAN: the launch covers games [span.first, span.end) only and files its leaves in batch span.batch
// (-1: batch by root mover's weight set, as above).  Two launches over the two halves of the slots, each
// with a batch and a stream of its own, let one half's tree step run under the other half's forward pass.
// HEADS = 1: from the linear layer's output (or its two split-K partial sums), head rows in LDS; HEADS = 2: from the
// per-column-tile partial sums of the heads that the split linear layer's last step leaves (128 bytes per leaf under
// the reference rule, 1 KB under PUCT; nothing staged in LDS).
template <bool AZ, int HEADS, bool TURNOVER, bool CACHE>
__global__ __launch_bounds__(kStepGames * 64) void tree_step_kernel(EngineDev e, SearchDev p, int parity, HeadsPair hp,
                                                                    RoundDev round, GameSpan span) {
    __shared__ int s_want[kStepGames];       // 0 = no leaf, 1 + batch otherwise
    __shared__ int s_base[2];
    extern __shared__ __attribute__((aligned(16))) float4 s_hw[];      // [batches][HROWS][256]
    constexpr int HROWS = AZ ? 8 : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = span.first + blockIdx.x * kStepGames + wv;
    if constexpr (HEADS == 1) {
        const int nb = p.wset[0] != p.wset[1] ? 2 : 1;
        for (int i = threadIdx.x; i < nb * HROWS * 256; i += kStepGames * 64) {
            const int b = i / (HROWS * 256), r = i % (HROWS * 256);
            s_hw[i] = reinterpret_cast<const float4 *>(hp.h[b].hw + (AZ ? 0 : 7 * 1024))[r];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {               // the next step's batch sizes start at zero
        if (span.batch < 0) { e.leaf_count[2 * (parity ^ 1)] = 0; e.leaf_count[2 * (parity ^ 1) + 1] = 0; }
        else e.leaf_count[2 * (parity ^ 1) + span.batch] = 0;
    }
    bool live = g < span.end && e.active[g];
    if constexpr (TURNOVER) {
        if (live && e.pending[g] < 0 && e.sims_done[g] >= p.sims) {
            if (lane == 0) {
                readout_game(e, p, g, nullptr, nullptr, nullptr, round, 1);
                if (e.active[g]) init_search(e, p, g, 0, 1);
            }
            wave_mem_sync();
            live = e.active[g];
        }
    }
    const bool two_batches = p.wset[0] != p.wset[1];
    int batch = 0;
    // leaf found by this wave (if any), kept in registers until the block has claimed its
    // slots in the compacted batch with ONE atomic (one per game serialised on the counter)
    bool want = false;
    u64 leaf_me = 0, leaf_opp = 0;
    unsigned leaf_legal_mask = 0;
    int leaf_len = 0, leaf_path = 0;
    int sims_done = 0, n_nodes = 0, depth_sum = 0, term_hits = 0, cache_hits = 0;
    if (live) {
    Node *pool = e.nodes + (size_t)g * e.pool_cap;
    float *prior = e.prior + (size_t)g * e.pool_cap;
    sims_done = e.sims_done[g];
    n_nodes = e.n_nodes[g];
    const int player = e.root_player[g];
    const u64 r1 = e.root_boards[2 * g], r2 = e.root_boards[2 * g + 1];
    // every leaf of a search is judged by the root mover's network; with one network, by the launch's batch
    batch = two_batches ? player - 1 : (span.batch < 0 ? 0 : span.batch);
    CacheEntry *table = nullptr;
    uint32_t gen = 0;
    if constexpr (CACHE) { table = e.cache + (size_t)g * e.cache_slots; gen = e.cache_gen[g]; }

    // masks of the leaves batched by the previous step (this step writes the other half)
    const uint8_t *legal_in = e.leaf_legal + (size_t)(parity ^ 1) * 2 * e.max_games;
    int pend = e.pending[g];
    if (pend >= 0) {
        int len = e.path_len[g];
        int mypath = e.path[g * kPathSlots + lane];
        double v;
        float pr = 0.0f;
        if constexpr (HEADS == 2) {
            const int b = pend >= e.max_games ? 1 : 0;
            const HeadsDev &hd = hp.h[b];
            const float *qrow = hd.q + (size_t)(pend - b * e.max_games) * 256;
            if constexpr (AZ) {
                float z[8], lp[7], pri[7];
                heads_z_from_q(qrow, hd.hb, lane, z);
                heads_log_softmax(z, lp);
                heads_priors(lp, legal_in[pend], pri);
#pragma unroll
                for (int o = 0; o < 7; ++o) pr = lane == o ? pri[o] : pr;
                v = (double)tanhf(z[7]);
            } else {
                v = (double)tanhf(heads_value_from_q(qrow, hd.hb, lane));   // value.item(), agent_code.py:213
            }
        } else if constexpr (HEADS == 1) {
            const int b = pend >= e.max_games ? 1 : 0;
            const HeadsDev &hd = hp.h[b];
            float4 xv[4];
            heads_load_row(hd, (size_t)(pend - b * e.max_games), lane, xv);
            const float4 *hw = s_hw + (two_batches ? b : 0) * HROWS * 256;
            if constexpr (AZ) {
                float z[8], lp[7], pri[7];
#pragma unroll
                for (int o = 0; o < 8; ++o) z[o] = heads_dot(xv, hw + o * 256, lane) + hd.hb[o];
                heads_log_softmax(z, lp);
                heads_priors(lp, legal_in[pend], pri);
#pragma unroll
                for (int o = 0; o < 7; ++o) pr = lane == o ? pri[o] : pr;
                v = (double)tanhf(z[7]);
            } else {
                v = (double)tanhf(heads_dot(xv, hw, lane) + hd.hb[7]);   // value.item(), agent_code.py:213
            }
        } else {
            v = (double)e.leaf_value[pend];              // value.item(), agent_code.py:213
            if (AZ) pr = lane < 7 ? e.leaf_prior[pend * 7 + lane] : 0.0f;
        }
        if constexpr (CACHE) {                            // file the evaluation under the leaf's position
            const u64 kme = e.pend_key[2 * g], kopp = e.pend_key[2 * g + 1];
            CacheEntry *ce = table + (c4::mix64(kme, kopp) & (u64)(e.cache_slots - 1));
            if (lane == 0) { ce->me = kme; ce->opp = kopp; ce->value = (float)v; ce->gen = gen; }
            if (AZ && lane < 7) ce->prior[lane] = pr;
        }
        if (AZ) {
            const int leaf = __shfl(mypath, len - 1);
            const unsigned legal = legal_in[pend];
            if (len == 1 && p.dirichlet_eps > 0.0 && lane < 7)
                pr = (float)((1.0 - p.dirichlet_eps) * (double)pr + p.dirichlet_eps * (double)e.root_noise[7 * g + lane]);
            n_nodes += expand_az(pool, prior, leaf, n_nodes, legal, pr, lane);
            backup_az(pool, mypath, len, v, lane);
        } else {
            backup_path(pool, mypath, len, v, lane);
        }
        wave_mem_sync();
        ++sims_done;
        pend = -1;
    }

    int spot = 0;
    while (sims_done < p.sims && spot < kTerminalSimsPerStep) {
        int mypath, len, node;
        u64 c1, c2;
        bool terminal;
        int mover = player;
        if (AZ) {
            DescentAz d = descend_az(pool, prior, p.c_puct, player, r1, r2, lane, mypath);
            c1 = d.c1; c2 = d.c2; len = d.len; node = d.node; terminal = d.terminal; mover = d.mover;
        } else {
            Descent d = descend_compat(pool, e.log_table, p.c_puct, player, r1, r2, lane, mypath);
            c1 = d.c1; c2 = d.c2; len = d.len; node = d.node; terminal = d.terminal;
        }
        depth_sum += len - 1;
        unsigned legal = c4::legal_mask(c1 | c2);
        if (terminal || legal == 0) {
            if (AZ) {
                double v = (c4::has_four(c1) || c4::has_four(c2)) ? -1.0 : 0.0;
                backup_az(pool, mypath, len, v, lane);
            } else {
                // agent_code.py:182-186 and :190-193
                double v = terminal ? c4::reward(c1, c2) : 0.0;
                backup_path(pool, mypath, len, v, lane);
            }
            wave_mem_sync();
            ++sims_done; ++term_hits; ++spot;
            continue;
        }
        // COMPAT: the leaf is encoded from the agent's side at every depth (agent_code.py:85;
        // `mover` is still the root mover here); AZ: from the side to move at the leaf
        const int view = AZ ? mover : view_side(e, p, g, player);
        leaf_me = view == 1 ? c1 : c2;
        leaf_opp = view == 1 ? c2 : c1;
        if constexpr (CACHE) {
            const u64 kopp = leaf_opp | ((u64)batch << 63);      // the two networks of an arena do not share entries
            const CacheEntry *ce = table + (c4::mix64(leaf_me, kopp) & (u64)(e.cache_slots - 1));
            if (ce->gen == gen && ce->me == leaf_me && ce->opp == kopp) {
                const double v = (double)ce->value;
                if (AZ) {
                    float pr = lane < 7 ? ce->prior[lane] : 0.0f;
                    if (len == 1 && p.dirichlet_eps > 0.0 && lane < 7)
                        pr = (float)((1.0 - p.dirichlet_eps) * (double)pr + p.dirichlet_eps * (double)e.root_noise[7 * g + lane]);
                    n_nodes += expand_az(pool, prior, node, n_nodes, legal, pr, lane);
                    backup_az(pool, mypath, len, v, lane);
                } else {
                    n_nodes += expand_compat(pool, node, n_nodes, legal, lane);
                    backup_path(pool, mypath, len, v, lane);
                }
                wave_mem_sync();
                ++sims_done; ++cache_hits; ++spot;
                continue;
            }
            if (lane == 0) { e.pend_key[2 * g] = leaf_me; e.pend_key[2 * g + 1] = kopp; }
        }
        if (!AZ) n_nodes += expand_compat(pool, node, n_nodes, legal, lane);
        want = true;
        leaf_legal_mask = legal;
        leaf_len = len;
        leaf_path = mypath;
        break;
    }
    }   // live

    if (lane == 0) s_want[wv] = want ? 1 + batch : 0;
    __syncthreads();
    if (threadIdx.x < 2) {
        int total = 0;
        for (int k = 0; k < kStepGames; ++k) total += s_want[k] == 1 + (int)threadIdx.x;
        s_base[threadIdx.x] = total ? atomicAdd(&e.leaf_count[2 * parity + threadIdx.x], total) : 0;
    }
    __syncthreads();
    if (!live) return;
    int pend_out = -1;
    if (want) {
        int slot = batch * e.max_games + s_base[batch];
        for (int k = 0; k < wv; ++k) slot += s_want[k] == 1 + batch;
        if (lane == 0) {
            e.leaf_side[2 * slot] = leaf_me;
            e.leaf_side[2 * slot + 1] = leaf_opp;
            e.leaf_legal[(size_t)parity * 2 * e.max_games + slot] = (uint8_t)leaf_legal_mask;
            e.path_len[g] = leaf_len;
        }
        e.path[g * kPathSlots + lane] = (uint16_t)leaf_path;
        pend_out = slot;
    }
    if (lane == 0) {
        e.sims_done[g] = sims_done;
        e.n_nodes[g] = n_nodes;
        e.pending[g] = pend_out;
        int32_t *st = e.game_stats + 8 * g;
        st[0] += depth_sum; st[1] += term_hits; st[2] += want ? 1 : 0; st[5] += cache_hits;
    }
}

// ---------------------------------------------------------------------------
// K2, fused form (fixed evaluators: CONST / HASH / ROLLOUT): the whole search of one move
// in one launch, the node pool resident in LDS.  One 64-thread block (one wavefront) per
// game.  Dynamic LDS: lds_nodes * 16 bytes of nodes, then (COMPAT) the log table of sims+2
// doubles, unless lds_logt is 0 and it is read from memory, or (AZ) one float prior per node.
// SPLIT = false: lds_nodes = 1 + 7*sims, the whole pool.  SPLIT = true: the search may create more
// nodes than its share of the LDS holds (a CU has 160 KB, and one wave per CU leaves three of its
// four SIMDs idle: 800 simulations need 112 KB); nodes lds_nodes and up then live in the game's
// pool in memory (SplitArray), so that several games share a CU whatever the simulation count.
// ---------------------------------------------------------------------------
template <bool AZ, bool SPLIT>
__global__ __launch_bounds__(64) void tree_fused_kernel(EngineDev e, SearchDev p, int lds_nodes, int lds_logt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    if (g >= e.n_games) return;
    if (!e.active[g]) return;
    Node *lpool = reinterpret_cast<Node *>(smem);
    double *llogt = reinterpret_cast<double *>(smem + (size_t)lds_nodes * sizeof(Node));
    float *lprior = reinterpret_cast<float *>(smem + (size_t)lds_nodes * sizeof(Node));
    const double *logt = e.log_table;
    if (!AZ && lds_logt) {
        for (int i = lane; i < p.sims + 2; i += 64) llogt[i] = e.log_table[i];
        logt = llogt;
    }
    Node *gpool = e.nodes + (size_t)g * e.pool_cap;
    using Pool = std::conditional_t<SPLIT, SplitArray<Node>, Node *>;
    using Priors = std::conditional_t<SPLIT, SplitArray<float>, float *>;
    Pool pool;
    Priors prior;
    if constexpr (SPLIT) {
        pool = {lpool, gpool, lds_nodes};
        prior = {lprior, e.prior + (size_t)g * e.pool_cap, lds_nodes};
    } else {
        pool = lpool;
        prior = lprior;
    }
    if (lane == 0) {
        Node r;
        r.value = 0.0; r.visits = AZ ? 0 : 1; r.first_child = 0; r.nchild = 0; r.action = 0xFF;
        pool[0] = r;
    }
    __syncthreads();

    const int player = e.root_player[g];
    const u64 r1 = e.root_boards[2 * g], r2 = e.root_boards[2 * g + 1];
    const uint32_t game = (uint32_t)e.game_id[g], move = (uint32_t)e.move_index[g];
    const u64 smix = set_mix(p.wset[player - 1]);
    const int view = view_side(e, p, g, player);
    int n_nodes = 1, depth_sum = 0, term_hits = 0, leaves = 0;
    for (int s = 0; s < p.sims; ++s) {
        int mypath;
        if (AZ) {
            DescentAz d = descend_az(pool, prior, p.c_puct, player, r1, r2, lane, mypath);
            depth_sum += d.len - 1;
            unsigned legal = c4::legal_mask(d.c1 | d.c2);
            double v;
            if (d.terminal || legal == 0) {
                v = (c4::has_four(d.c1) || c4::has_four(d.c2)) ? -1.0 : 0.0;
                ++term_hits;
            } else {
                const u64 me = d.mover == 1 ? d.c1 : d.c2, opp = d.mover == 1 ? d.c2 : d.c1;
                float pr = lane < 7 ? fixed_prior(p.evaluator, me ^ smix, opp, legal, lane) : 0.0f;
                if (d.len == 1 && p.dirichlet_eps > 0.0 && lane < 7)
                    pr = (float)((1.0 - p.dirichlet_eps) * (double)pr + p.dirichlet_eps * (double)e.root_noise[7 * g + lane]);
                n_nodes += expand_az(pool, prior, d.node, n_nodes, legal, pr, lane);
                v = fixed_eval_az(p, d.c1, d.c2, d.mover, game, move, (uint32_t)s, smix);
                ++leaves;
            }
            backup_az(pool, mypath, d.len, v, lane);
        } else {
            Descent d = descend_compat(pool, logt, p.c_puct, player, r1, r2, lane, mypath);
            depth_sum += d.len - 1;
            unsigned legal = c4::legal_mask(d.c1 | d.c2);
            double v;
            if (d.terminal || legal == 0) {
                v = d.terminal ? c4::reward(d.c1, d.c2) : 0.0;
                ++term_hits;
            } else {
                n_nodes += expand_compat(pool, d.node, n_nodes, legal, lane);
                if (p.evaluator == NNB_EVAL_ROLLOUT) {
                    // playout value in get_reward()'s absolute frame, root mover moving first
                    int res = rollout(d.c1, d.c2, player, p.seed, game, move, (uint32_t)s);
                    v = res == 1 ? -1.0 : (res == -1 ? 1.0 : 0.0);
                } else {
                    v = fixed_eval(p.evaluator, (view == 1 ? d.c1 : d.c2) ^ smix, view == 1 ? d.c2 : d.c1);
                }
                ++leaves;
            }
            backup_path(pool, mypath, d.len, v, lane);
        }
        __syncthreads();
    }
    // hand the root and its children to the readout kernel through the global pool
    if (lane < 8 && lane < n_nodes) gpool[lane] = lpool[lane];
    if (lane == 0) {
        e.sims_done[g] = p.sims;
        e.n_nodes[g] = n_nodes;
        e.pending[g] = -1;
        int32_t *st = e.game_stats + 8 * g;
        st[0] += depth_sum; st[1] += term_hits; st[2] += leaves;
    }
}

// ---------------------------------------------------------------------------
// K7: end of a move.  Root readout (agent_code.py:217-229: most visited child, first
// maximum = lowest column), then for self-play the move is applied, the ply is
// appended to the game's trajectory and finished games are retired
// (train_alpha_zero.py:41-48).  One thread per game.
// ---------------------------------------------------------------------------
// Puts game j of the round into slot g; returns false when that game starts from a finished
// position (its record is then complete and the slot is still free).
// A game's record is complete: count it and, if the caller keeps one, append it to the completion list.
__device__ void note_finished(const EngineDev &e, const RoundDev &r, int record) {
    const unsigned k = (unsigned)atomicAdd(&e.counters[kFinished], 1);
    if (r.finished_ring) r.finished_ring[k % (unsigned)r.ring_len] = record;
}

__device__ bool seat_game(const EngineDev &e, const RoundDev &r, int g, int j) {
    // caller-supplied positions are reduced to something the kernels can index safely: only
    // the 42 playable bits, no cell owned twice, mover 1 or 2
    const int js = j % r.n_starts, jt = j % r.ring_len;
    u64 b1 = r.start_boards ? r.start_boards[2 * js] & c4::kBoardMask : 0;
    u64 b2 = r.start_boards ? r.start_boards[2 * js + 1] & c4::kBoardMask & ~b1 : 0;
    int player = r.start_players ? (r.start_players[js] == 2 ? 2 : 1) : 1;
    nnb_trajectory *t = r.traj + jt;
    t->length = 0; t->result = 2; t->game_id = r.first_game_id + j; t->reserved = 0;
    t->final_boards[0] = b1; t->final_boards[1] = b2;
    uint4 *z = reinterpret_cast<uint4 *>(t->plies);           // plies past `length` read as zero
    for (int k = 0; k < (int)(sizeof(t->plies) / (sizeof(uint4))); ++k) z[k] = make_uint4(0, 0, 0, 0);
    if (c4::is_terminal(b1, b2)) {
        t->result = c4::result(b1, b2);
        note_finished(e, r, jt);
        return false;
    }
    e.root_boards[2 * g] = b1; e.root_boards[2 * g + 1] = b2;
    e.root_player[g] = (int8_t)player;
    e.root_side[g] = 0;                                       // self-play: SearchDev::side decides
    e.ply[g] = 0;
    e.game_id[g] = r.first_game_id + j;
    e.traj_index[g] = jt;
    e.cache_gen[g] += 1;                                      // the new game starts with an empty evaluation cache
    return true;
}

__device__ void readout_game(const EngineDev &e, const SearchDev &p, int g, int32_t *__restrict__ visits_out,
                             int8_t *__restrict__ action_out, double *__restrict__ root_value_out,
                             const RoundDev &round, int selfplay) {
    const Node *pool = e.nodes + (size_t)g * e.pool_cap;
    Node root = pool[0];
    int vis[7] = {0, 0, 0, 0, 0, 0, 0};
    int best_a = -1, best_v = -1;
    for (int k = 0; k < root.nchild; ++k) {
        Node ch = pool[root.first_child + k];
        vis[ch.action] = ch.visits;
        if (ch.visits > best_v) { best_v = ch.visits; best_a = ch.action; }
    }
    if (visits_out)
        for (int c = 0; c < 7; ++c) visits_out[7 * g + c] = vis[c];
    if (action_out) action_out[g] = (int8_t)best_a;
    if (root_value_out) root_value_out[g] = root.value;
    e.game_stats[8 * g + 3] += e.sims_done[g];
    if (!selfplay) return;

    u64 b1 = e.root_boards[2 * g], b2 = e.root_boards[2 * g + 1];
    int player = e.root_player[g];
    int ply = e.ply[g];
    nnb_trajectory *t = round.traj + e.traj_index[g];
    bool done = best_a < 0 || ply >= kMaxPlies;
    if (!done) {
        nnb_ply *rec = &t->plies[ply];
        rec->colour_boards[0] = b1; rec->colour_boards[1] = b2;
        for (int c = 0; c < 7; ++c) rec->visits[c] = (uint16_t)vis[c];
        rec->action = (uint8_t)best_a;
        rec->player = (uint8_t)player;
        u64 cell = c4::drop_cell(b1 | b2, best_a);            // game.make_move, train_alpha_zero.py:44
        if (player == 1) b1 |= cell; else b2 |= cell;
        player = 3 - player;                                  // connect_four_game.py:50
        ++ply;
        e.root_boards[2 * g] = b1; e.root_boards[2 * g + 1] = b2;
        e.root_player[g] = (int8_t)player;
        e.ply[g] = ply;
        done = c4::is_terminal(b1, b2);                       // train_alpha_zero.py:41
    }
    if (done) {
        t->length = ply;
        t->result = c4::result(b1, b2);                       // train_alpha_zero.py:48
        t->final_boards[0] = b1; t->final_boards[1] = b2;
        e.game_stats[8 * g + 4] += ply;
        e.ply[g] = 0;
        __threadfence();                                          // the record is complete before it is listed
        note_finished(e, round, e.traj_index[g]);
        // the slot takes the next unplayed game of the round, if any
        bool seated = false;
        int tries = 0;
        while (!seated) {
            int j = e.n_games + atomicAdd(&e.counters[kRefills], 1);
            // (an endless stream whose start positions are all finished must not keep this lane here for ever)
            if (j >= round.total_games || (round.total_games >= kEndlessGames && ++tries > kSeatTries)) break;
            seated = seat_game(e, round, g, j);
        }
        if (!seated) { e.active[g] = 0; atomicSub(&e.counters[kLive], 1); }
        done = !seated;
    }
    if (!done) atomicAdd(&e.counters[0], 1);
}

__global__ void readout_kernel(EngineDev e, SearchDev p, int32_t *__restrict__ visits_out,
                               int8_t *__restrict__ action_out, double *__restrict__ root_value_out,
                               RoundDev round, int selfplay) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n_games) return;
    if (!e.active[g]) return;
    readout_game(e, p, g, visits_out, action_out, root_value_out, round, selfplay);
}

// Self-play without a common move boundary: launched before every tree step, it ends the move
// of exactly those games whose simulation budget is used up (root readout, move, trajectory,
// retirement or the next game of the round) and starts their next search.  Games that search
// long positions no longer hold back games that are done, so the leaf batch stays full.
__global__ void turnover_kernel(EngineDev e, SearchDev p, RoundDev round, int make_noise, GameSpan span) {
    int g = span.first + blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= span.end) return;
    if (!e.active[g]) return;
    if (e.pending[g] >= 0 || e.sims_done[g] < p.sims) return;
    readout_game(e, p, g, nullptr, nullptr, nullptr, round, 1);
    if (e.active[g]) init_search(e, p, g, make_noise, 1);
}

// Self-play set-up: slot g starts with game g of the round - the empty board
// (ConnectFourGame.reset, connect_four_game.py:30-36) or the caller's start position.  A game
// that starts from a finished position is retired at once and the slot takes another one.
__global__ void selfplay_init_kernel(EngineDev e, RoundDev round) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n_games) return;
    for (int k = 0; k < 8; ++k) e.game_stats[8 * g + k] = 0;
    e.ply[g] = 0;
    bool seated = seat_game(e, round, g, g);
    int tries = 0;
    while (!seated) {
        int j = e.n_games + atomicAdd(&e.counters[kRefills], 1);
        if (j >= round.total_games || (round.total_games >= kEndlessGames && ++tries > kSeatTries)) break;
        seated = seat_game(e, round, g, j);
    }
    e.active[g] = seated ? 1 : 0;
    if (seated) { atomicAdd(&e.counters[0], 1); atomicAdd(&e.counters[kLive], 1); }
}

// Search-only set-up: caller-supplied roots, every root searched.
__global__ void search_init_kernel(EngineDev e, const u64 *__restrict__ boards,
                                   const int8_t *__restrict__ players, const int8_t *__restrict__ sides,
                                   int game_base, const float *__restrict__ root_noise) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n_games) return;
    const u64 b1 = boards[2 * g] & c4::kBoardMask, b2 = boards[2 * g + 1] & c4::kBoardMask & ~b1;
    e.root_boards[2 * g] = b1; e.root_boards[2 * g + 1] = b2;
    e.root_player[g] = players[g] == 2 ? 2 : 1;
    e.root_side[g] = sides ? (sides[g] == 2 ? 2 : 1) : 0;
    e.cache_gen[g] += 1;
    e.active[g] = 1;
    e.ply[g] = 0;
    e.game_id[g] = game_base + g;
    if (root_noise)
        for (int c = 0; c < 7; ++c) e.root_noise[7 * g + c] = root_noise[7 * g + c];
    for (int k = 0; k < 8; ++k) e.game_stats[8 * g + k] = 0;
}

// Sums the per-game statistics into counters[8..11] (depth, terminal hits, leaves, sims)
// and counters[12] (plies).
__global__ void stats_reduce_kernel(EngineDev e) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    long long v[6] = {0, 0, 0, 0, 0, 0};
    if (g < e.n_games) {
        for (int k = 0; k < 4; ++k) v[k] = e.game_stats[8 * g + k];
        v[4] = e.game_stats[8 * g + 4] + e.ply[g];
        v[5] = e.game_stats[8 * g + 5];
    }
    for (int k = 0; k < 6; ++k) {
        long long x = v[k];
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x)
            atomicAdd(reinterpret_cast<unsigned long long *>(e.stats64) + k, (unsigned long long)x);
    }
}

// Fixed evaluators over the compacted leaf batch: fills the value buffer exactly as the
// network's heads kernel would, so the lock-step path can be run without weights.
__global__ void fixed_eval_kernel(EngineDev e, SearchDev p, int batch, int parity) {   // one launch per batch
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.leaf_count[2 * parity + batch]) return;
    i += batch * e.max_games;
    const u64 me = e.leaf_side[2 * i] ^ set_mix(p.wset[batch]), opp = e.leaf_side[2 * i + 1];
    e.leaf_value[i] = (float)fixed_eval(p.evaluator, me, opp);
    if (p.mode == NNB_MODE_AZ) {
        const unsigned legal = e.leaf_legal[(size_t)parity * 2 * e.max_games + i];
        for (int c = 0; c < 7; ++c) e.leaf_prior[7 * i + c] = fixed_prior(p.evaluator, me, opp, legal, c);
    }
}

// ---------------------------------------------------------------------------
// Depth-limited minimax opponent (nnbattle/agents/minimax/agent_code.py:34-89) with the position
// score of connect_four_game.py:215-278.  The maximising side is the side to move (the reference
// always maximises for colour 2; the two agree whenever colour 2 moves).  Alpha-beta only prunes,
// it never changes the root's choice, so the tree is searched exhaustively here: one wavefront per
// position, lane 7*c1 + c2 owns the subtree behind the first two plies (c1, c2), the wave reduces
// min over c2 then max over c1, lowest column on ties (the reference keeps the first maximum).
// ---------------------------------------------------------------------------
constexpr int kMinimaxInf = 1 << 30;
constexpr int kMinimaxMaxDepth = 8;

struct WindowTable { u64 m[69]; };
__host__ __device__ constexpr u64 window_mask(int r0, int c0, int dr, int dc) {    // rows counted from the top
    u64 m = 0;
    for (int i = 0; i < 4; ++i) m |= 1ULL << ((c0 + dc * i) * 7 + (5 - (r0 + dr * i)));
    return m;
}
__host__ __device__ constexpr WindowTable make_windows() {
    WindowTable t{};
    int n = 0;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 4; ++c) t.m[n++] = window_mask(r, c, 0, 1);
    for (int c = 0; c < 7; ++c) for (int r = 0; r < 3; ++r) t.m[n++] = window_mask(r, c, 1, 0);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) t.m[n++] = window_mask(r, c, 1, 1);
    for (int r = 3; r < 6; ++r) for (int c = 0; c < 4; ++c) t.m[n++] = window_mask(r, c, -1, 1);
    return t;
}
__constant__ WindowTable c_windows = make_windows();

// score_position / evaluate_window (connect_four_game.py:215-278) for the side owning `me`
__device__ int score_position(u64 me, u64 opp) {
    const u64 empty = ~(me | opp);
    int score = 3 * __popcll(me & (0x3FULL << 21));
    for (int i = 0; i < 69; ++i) {
        const u64 w = c_windows.m[i];
        const int a = __popcll(me & w), o = __popcll(opp & w), e = __popcll(empty & w);
        score += a == 4 ? 100 : (a == 3 && e == 1) ? 5 : (a == 2 && e == 2) ? 2 : 0;
        score -= (o == 3 && e == 1) ? 4 : 0;
    }
    return score;
}

// Exhaustive minimax value of (mine, theirs) for the maximising side `mine`; `my_turn` = the
// maximising side is to move.  Iterative depth-first walk, `depth` plies deep.
__device__ int minimax_value(u64 mine, u64 theirs, bool my_turn, int depth) {
    u64 sm[kMinimaxMaxDepth + 1], st[kMinimaxMaxDepth + 1];
    int val[kMinimaxMaxDepth + 1], col[kMinimaxMaxDepth + 1];
    int L = 0;
    sm[0] = mine; st[0] = theirs; col[0] = -1;
    int ret = 0;
    bool have_ret = false;
    while (true) {
        const bool maxi = (L & 1) ? !my_turn : my_turn;
        if (col[L] < 0) {                                   // first visit of this node
            if (c4::has_four(sm[L])) { ret = kMinimaxInf; have_ret = true; }
            else if (c4::has_four(st[L])) { ret = -kMinimaxInf; have_ret = true; }
            else if (c4::legal_mask(sm[L] | st[L]) == 0) { ret = 0; have_ret = true; }
            else if (L == depth) { ret = score_position(sm[L], st[L]); have_ret = true; }
            else { val[L] = maxi ? -kMinimaxInf - 1 : kMinimaxInf + 1; col[L] = 0; }
        }
        if (!have_ret) {
            const unsigned legal = c4::legal_mask(sm[L] | st[L]);
            int c = col[L];
            while (c < 7 && !((legal >> c) & 1)) ++c;
            if (c >= 7) { ret = val[L]; have_ret = true; }
            else {
                col[L] = c + 1;
                const u64 cell = c4::drop_cell(sm[L] | st[L], c);
                sm[L + 1] = maxi ? sm[L] | cell : sm[L];
                st[L + 1] = maxi ? st[L] : st[L] | cell;
                col[L + 1] = -1;
                ++L;
                continue;
            }
        }
        // hand `ret` to the parent
        if (L == 0) return ret;
        --L;
        have_ret = false;
        const bool pmax = (L & 1) ? !my_turn : my_turn;
        val[L] = pmax ? (ret > val[L] ? ret : val[L]) : (ret < val[L] ? ret : val[L]);
    }
}

__global__ __launch_bounds__(64) void minimax_kernel(int n, const u64 *__restrict__ boards,
                                                     const int8_t *__restrict__ movers, int depth,
                                                     int8_t *__restrict__ col_out, int32_t *__restrict__ score_out) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= n) return;
    const u64 b1 = boards[2 * g] & c4::kBoardMask, b2 = boards[2 * g + 1] & c4::kBoardMask & ~b1;
    const bool two = movers[g] == 2;
    const u64 mine = two ? b2 : b1, theirs = two ? b1 : b2;
    const unsigned legal = c4::legal_mask(mine | theirs);
    int root_col = -1, root_val;
    if (c4::has_four(mine)) root_val = kMinimaxInf;
    else if (c4::has_four(theirs)) root_val = -kMinimaxInf;
    else if (legal == 0) root_val = 0;
    else if (depth == 0) root_val = score_position(mine, theirs);
    else {
        // lane -> (c1, c2); value of the position behind those plies from the maximiser's side
        const int c1 = lane / 7, c2 = lane % 7;
        int v = kMinimaxInf + 1;                             // neutral for the min over c2
        bool c1_final = false;                               // c1 ends the search (win, full board, horizon)
        int v1_direct = 0;
        if (lane < 49 && ((legal >> c1) & 1)) {
            const u64 m1 = mine | c4::drop_cell(mine | theirs, c1);
            const unsigned legal1 = c4::legal_mask(m1 | theirs);
            if (c4::has_four(m1)) { c1_final = true; v1_direct = kMinimaxInf; }
            else if (legal1 == 0) { c1_final = true; v1_direct = 0; }
            else if (depth == 1) { c1_final = true; v1_direct = score_position(m1, theirs); }
            else if ((legal1 >> c2) & 1) {
                const u64 t2 = theirs | c4::drop_cell(m1 | theirs, c2);
                v = minimax_value(m1, t2, true, depth - 2);
            }
        }
        // min over c2 inside each group of 7 lanes, then max over c1 (lowest column wins ties)
        int best = -kMinimaxInf - 2;
        for (int k = 0; k < 7; ++k) {
            int vk = kMinimaxInf + 1;
            for (int j = 0; j < 7; ++j) { const int x = __shfl(v, 7 * k + j); vk = x < vk ? x : vk; }
            const int fin = __shfl((int)c1_final, 7 * k), dir = __shfl(v1_direct, 7 * k);
            if (fin) vk = dir;
            if (((legal >> k) & 1) && vk > best) { best = vk; root_col = k; }
        }
        root_val = best;
    }
    if (lane == 0) {
        if (col_out) col_out[g] = (int8_t)root_col;
        if (score_out) score_out[g] = root_val;
    }
}

// ---------------------------------------------------------------------------
// Host launchers
// ---------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void launch_game_ops(hipStream_t st, int n, const u64 *in, const int8_t *players, const int8_t *actions,
                     u64 *out, uint8_t *legal_out, uint8_t *status_out) {
    if (n <= 0) return;
    game_ops_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, in, players, actions, out, legal_out, status_out);
}

void launch_encode_planes(hipStream_t st, int n, const u64 *side, float *planes) {
    if (n <= 0) return;
    encode_planes_kernel<<<cdiv(n * 84, 256), 256, 0, st>>>(n, side, planes);
}

void launch_trajectory_samples(hipStream_t st, int n, const nnb_trajectory *rec, const int32_t *ply_offsets, float *planes,
                               float *probs, float *z) {
    if (n <= 0) return;
    trajectory_samples_kernel<<<cdiv(n * kMaxPlies, 256), 256, 0, st>>>(n, rec, ply_offsets, planes, probs, z);
}

void launch_ucb_scores(hipStream_t st, int n, const double *value, const int32_t *visits,
                       const int32_t *parent_visits, double c_puct, const double *log_table, int log_n,
                       double *out) {
    if (n <= 0) return;
    ucb_scores_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, value, visits, parent_visits, c_puct, log_table,
                                                    log_n, out);
}

void launch_tree_init(hipStream_t st, const EngineDev &e, const SearchDev &p, int make_noise, int selfplay) {
    tree_init_kernel<<<cdiv(e.n_games, 256), 256, 0, st>>>(e, p, make_noise, selfplay);
}

template <bool AZ, int HEADS>
static void launch_tree_step_t(hipStream_t st, const EngineDev &e, const SearchDev &p, int parity, const HeadsPair &heads,
                               const RoundDev *turnover, size_t lds, const GameSpan &span) {
    const int blocks = cdiv(span.end - span.first, kStepGames), threads = kStepGames * 64;
    if (blocks <= 0) return;
    const RoundDev none = {nullptr, nullptr, nullptr, 0, 0, 1, 1, nullptr};
    const bool cache = (p.flags & NNB_FLAG_EVAL_CACHE) && e.cache;
    if (turnover) {
        if (cache) tree_step_kernel<AZ, HEADS, true, true><<<blocks, threads, lds, st>>>(e, p, parity, heads, *turnover, span);
        else tree_step_kernel<AZ, HEADS, true, false><<<blocks, threads, lds, st>>>(e, p, parity, heads, *turnover, span);
    } else {
        if (cache) tree_step_kernel<AZ, HEADS, false, true><<<blocks, threads, lds, st>>>(e, p, parity, heads, none, span);
        else tree_step_kernel<AZ, HEADS, false, false><<<blocks, threads, lds, st>>>(e, p, parity, heads, none, span);
    }
}

// turnover != null: games whose budget is used up end their move inside the step (see TURNOVER)
void launch_tree_step(hipStream_t st, const EngineDev &e, const SearchDev &p, int parity, const HeadsPair &heads,
                      const RoundDev *turnover, const GameSpan *span_or_null) {
    const bool az = p.mode == NNB_MODE_AZ;
    const GameSpan span = span_or_null ? *span_or_null : GameSpan{0, e.n_games, -1};
    if (heads.fused && heads.h[0].q) {
        if (az) launch_tree_step_t<true, 2>(st, e, p, parity, heads, turnover, 0, span);
        else launch_tree_step_t<false, 2>(st, e, p, parity, heads, turnover, 0, span);
    } else if (heads.fused) {
        const size_t lds = (size_t)(p.wset[0] != p.wset[1] ? 2 : 1) * (az ? 8 : 1) * 4096;
        if (az) launch_tree_step_t<true, 1>(st, e, p, parity, heads, turnover, lds, span);
        else launch_tree_step_t<false, 1>(st, e, p, parity, heads, turnover, lds, span);
    } else {
        if (az) launch_tree_step_t<true, 0>(st, e, p, parity, heads, turnover, 0, span);
        else launch_tree_step_t<false, 0>(st, e, p, parity, heads, turnover, 0, span);
    }
}

hipError_t prepare_tree_fused(size_t lds_bytes) {
    const void *k[4] = {reinterpret_cast<const void *>(tree_fused_kernel<false, false>),
                        reinterpret_cast<const void *>(tree_fused_kernel<true, false>),
                        reinterpret_cast<const void *>(tree_fused_kernel<false, true>),
                        reinterpret_cast<const void *>(tree_fused_kernel<true, true>)};
    for (const void *f : k) {
        hipError_t err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

// The PUCT step kernel of a two-network arena stages 2 x 8 head rows = 64 KB of dynamic LDS.
hipError_t prepare_tree_step() {
    const void *k[4] = {reinterpret_cast<const void *>(tree_step_kernel<true, 1, false, false>),
                        reinterpret_cast<const void *>(tree_step_kernel<true, 1, true, false>),
                        reinterpret_cast<const void *>(tree_step_kernel<true, 1, false, true>),
                        reinterpret_cast<const void *>(tree_step_kernel<true, 1, true, true>)};
    for (const void *f : k) {
        hipError_t err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4096);
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

FusedLds fused_lds_plan(int sims, bool az, size_t budget) {
    FusedLds f;
    const size_t pool_nodes = 1 + 7 * (size_t)sims;
    const size_t logt_bytes = az ? 0 : (size_t)(sims + 2) * sizeof(double);
    const size_t per_node = sizeof(Node) + (az ? sizeof(float) : 0);
    f.nodes = (int)pool_nodes;
    f.logt = !az;
    f.bytes = pool_nodes * per_node + logt_bytes;
    if (f.bytes > budget) {
        // a log table that would take more than a quarter of the share is read from memory instead
        f.logt = !az && logt_bytes * 4 <= budget;
        const size_t left = budget - (f.logt ? logt_bytes : 0);
        f.nodes = (int)(left / per_node);
        if (f.nodes > (int)pool_nodes) f.nodes = (int)pool_nodes;
        f.bytes = (size_t)f.nodes * per_node + (f.logt ? logt_bytes : 0);
    }
    f.split = f.nodes < (int)pool_nodes;
    return f;
}

void launch_tree_fused(hipStream_t st, const EngineDev &e, const SearchDev &p, const FusedLds &f) {
    const bool az = p.mode == NNB_MODE_AZ;
    if (f.split) {
        if (az) tree_fused_kernel<true, true><<<e.n_games, 64, f.bytes, st>>>(e, p, f.nodes, f.logt);
        else tree_fused_kernel<false, true><<<e.n_games, 64, f.bytes, st>>>(e, p, f.nodes, f.logt);
    } else {
        if (az) tree_fused_kernel<true, false><<<e.n_games, 64, f.bytes, st>>>(e, p, f.nodes, f.logt);
        else tree_fused_kernel<false, false><<<e.n_games, 64, f.bytes, st>>>(e, p, f.nodes, f.logt);
    }
}

void launch_dirichlet_noise(hipStream_t st, int n, u64 seed, const int32_t *game_ids, const int32_t *move_index,
                            const uint8_t *legal, double alpha, float *out) {
    if (n <= 0) return;
    dirichlet_noise_kernel<<<cdiv(n, 64), 64, 0, st>>>(n, seed, game_ids, move_index, legal, alpha, out);
}

void launch_readout(hipStream_t st, const EngineDev &e, const SearchDev &p, int32_t *visits_out,
                    int8_t *action_out, double *root_value_out, const RoundDev *round) {
    RoundDev none = {nullptr, nullptr, nullptr, 0, 0, 1, 1, nullptr};
    readout_kernel<<<cdiv(e.n_games, 256), 256, 0, st>>>(e, p, visits_out, action_out, root_value_out,
                                                         round ? *round : none, round != nullptr);
}

void launch_turnover(hipStream_t st, const EngineDev &e, const SearchDev &p, const RoundDev &round, int make_noise,
                     const GameSpan *span_or_null) {
    const GameSpan span = span_or_null ? *span_or_null : GameSpan{0, e.n_games, -1};
    if (span.end <= span.first) return;
    turnover_kernel<<<cdiv(span.end - span.first, 256), 256, 0, st>>>(e, p, round, make_noise, span);
}

void launch_minimax(hipStream_t st, int n, const u64 *boards, const int8_t *movers, int depth, int8_t *col_out,
                    int32_t *score_out) {
    minimax_kernel<<<n, 64, 0, st>>>(n, boards, movers, depth, col_out, score_out);
}

void launch_selfplay_init(hipStream_t st, const EngineDev &e, const RoundDev &round) {
    selfplay_init_kernel<<<cdiv(e.n_games, 256), 256, 0, st>>>(e, round);
}

void launch_search_init(hipStream_t st, const EngineDev &e, const u64 *boards, const int8_t *players,
                        const int8_t *sides, int game_base, const float *root_noise) {
    search_init_kernel<<<cdiv(e.n_games, 256), 256, 0, st>>>(e, boards, players, sides, game_base, root_noise);
}

void launch_fixed_eval(hipStream_t st, const EngineDev &e, const SearchDev &p, int batch, int parity) {
    fixed_eval_kernel<<<cdiv(e.n_games, 256), 256, 0, st>>>(e, p, batch, parity);
}

void launch_stats_reduce(hipStream_t st, const EngineDev &e) {
    stats_reduce_kernel<<<cdiv(e.n_games, 256), 256, 0, st>>>(e);
}

}  // namespace nnb

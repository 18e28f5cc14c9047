/*
 * Fast CPU restatement (plain C, bitboards) of the nnbattle Connect Four rules
 * and the reference's MCTS search.
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, the smoke check and the benchmark's
 * CPU-baseline leg; never by the shipped library.  Parity is pinned: it is
 * checked against vectors captured from the imported reference
 * (tests/golden/, written by oracle/capture_golden.py).
 *
 * Bit layout: bit index = col*7 + row_from_bottom (row_from_bottom = 5 - the
 * reference's row index, whose row 0 is the top: connect_four_game.py:21,47).
 * Bit 6 of every 7-bit column stays zero, so shifted ANDs cannot wrap.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the selection score must round exactly like the
 * CPython expression at mcts.py:29.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;

#define ORC_COLS 7
#define ORC_ROWS 6
#define ORC_FULL_MASK 0x0FDFBF7EFDFBFULL /* 6 low bits of each of 7 columns */

enum { ORC_EVAL_CONST = 0, ORC_EVAL_HASH_Q8 = 1, ORC_EVAL_HASH_F32 = 2, ORC_EVAL_CALLBACK = 3, ORC_EVAL_ROLLOUT = 4 };

int orc_rollout(u64 b1, u64 b2, int mover, u64 seed, uint32_t game_id, uint32_t move_index, uint32_t sim);

/* value callback: (ctx, me_bits, opp_bits) -> value already widened to double */
typedef double (*orc_eval_fn)(void *ctx, u64 me, u64 opp);

/* ---- rules ------------------------------------------------------------ */

/* connect_four_game.py:117-125 - column open iff its top cell (row 0) is empty */
int orc_legal_mask(u64 b1, u64 b2) {
    u64 occ = b1 | b2;
    int m = 0;
    for (int c = 0; c < ORC_COLS; ++c)
        if (!((occ >> (c * 7 + 5)) & 1)) m |= 1 << c;
    return m;
}

/* connect_four_game.py:38-52 - drop in the lowest empty cell; 0 if the column is full */
int orc_drop(u64 *b1, u64 *b2, int player, int col) {
    u64 occ = *b1 | *b2;
    u64 bit = 0;
    if ((occ >> (col * 7 + 5)) & 1) return 0;
    /* columns fill bottom-up during play, but callers may hand in arbitrary boards:
       scan like the reference does (rows 5 -> 0 == bottom -> top). */
    for (int r = 0; r < ORC_ROWS; ++r) {
        u64 b = 1ULL << (col * 7 + r);
        if (!(occ & b)) { bit = b; break; }
    }
    if (player == 1) *b1 |= bit; else *b2 |= bit;
    return 1;
}

/* connect_four_game.py:160-192 - any of the 69 windows filled by one colour */
int orc_has_four(u64 b) {
    u64 m;
    m = b & (b >> 1);  if (m & (m >> 2))  return 1;  /* vertical   */
    m = b & (b >> 7);  if (m & (m >> 14)) return 1;  /* horizontal */
    m = b & (b >> 6);  if (m & (m >> 12)) return 1;  /* diagonal   */
    m = b & (b >> 8);  if (m & (m >> 16)) return 1;  /* diagonal   */
    return 0;
}

/* connect_four_game.py:194-200 */
int orc_is_full(u64 b1, u64 b2) { return ((b1 | b2) & ORC_FULL_MASK) == ORC_FULL_MASK; }

/* connect_four_game.py:72-78 */
int orc_is_terminal(u64 b1, u64 b2) { return orc_has_four(b1) || orc_has_four(b2) || orc_is_full(b1, b2); }

/* connect_four_game.py:80-91 - colour 2 is checked first */
double orc_reward(u64 b1, u64 b2) {
    if (orc_has_four(b2)) return 1.0;
    if (orc_has_four(b1)) return -1.0;
    return 0.0;
}

/* connect_four_game.py:202-213 */
int orc_winner(u64 b1, u64 b2) {
    if (orc_has_four(b2)) return 2;
    if (orc_has_four(b1)) return 1;
    return 0;
}

/* connect_four_game.py:280-296 - colour 1 is checked first; 2 encodes None */
int orc_result(u64 b1, u64 b2) {
    if (!orc_is_terminal(b1, b2)) return 2;
    if (orc_has_four(b1)) return 1;
    if (orc_has_four(b2)) return -1;
    return 0;
}

/* ---- fixed evaluators ------------------------------------------------- */
u64 orc_mix64(u64 me, u64 opp) {
    u64 x = me ^ (opp * 0x9E3779B97F4A7C15ULL);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

double orc_hash_value_q8(u64 me, u64 opp) {
    return ((double)(int64_t)(orc_mix64(me, opp) >> 55) - 256.0) / 256.0;
}

double orc_hash_value_f32(u64 me, u64 opp) {
    u64 x = orc_mix64(me, opp);
    double v = ((double)(x >> 40) / 8388608.0 - 1.0) / (double)(1u << ((x >> 37) & 7));
    return (double)(float)v;
}

static double eval_leaf(int kind, orc_eval_fn fn, void *ctx, u64 me, u64 opp) {
    switch (kind) {
    case ORC_EVAL_CONST:    return 0.0;
    case ORC_EVAL_HASH_Q8:  return orc_hash_value_q8(me, opp);
    case ORC_EVAL_HASH_F32: return orc_hash_value_f32(me, opp);
    default:                return fn(ctx, me, opp);
    }
}

/* ---- search ----------------------------------------------------------- */
typedef struct {
    int64_t nodes;        /* nodes allocated, root included          */
    int64_t depth_sum;    /* sum over simulations of selection depth */
    int64_t evals;        /* evaluator calls (non-terminal leaves)   */
    int64_t terminal_hits;
    int32_t root_visits;
    double  root_value;
} orc_stats;

/*
 * agent_code.py:144-234 with mcts.py:13-44.  Kept quirks (each is a line of the
 * reference): root.visits = 1 (:160); descent places the root mover's colour at
 * every depth (connect_four_game.py:50 then agent_code.py:176); the leaf is
 * encoded from agent.current_player's side (:85-87; `side`, the mover in self-play); absolute +-1 terminal reward
 * (:182-186) with no negation in the backup (mcts.py:40-44); priors discarded
 * (mcts.py:35-38); lowest column wins ties (mcts.py:30, agent_code.py:223).
 * Returns the selected column, or -1 when the root never got children.
 */
static int compat_search_impl(u64 b1, u64 b2, int player, int side, int sims, double c_puct, int eval_kind,
                              orc_eval_fn fn, void *ctx, u64 seed, uint32_t game_id, uint32_t move_index,
                              int32_t *visits_out, orc_stats *st) {
    int cap = 1 + 7 * sims;
    int32_t *visits = (int32_t *)calloc(cap, sizeof(int32_t));
    double *value = (double *)calloc(cap, sizeof(double));
    int32_t *parent = (int32_t *)malloc(cap * sizeof(int32_t));
    int32_t *first = (int32_t *)malloc(cap * sizeof(int32_t));
    int8_t *nchild = (int8_t *)calloc(cap, 1);
    int8_t *action = (int8_t *)malloc(cap);
    int n_nodes = 1;
    int64_t depth_sum = 0, evals = 0, term = 0;
    visits[0] = 1; parent[0] = -1; first[0] = -1; action[0] = -1;

    for (int s = 0; s < sims; ++s) {
        int node = 0;
        u64 c1 = b1, c2 = b2;
        while (nchild[node] > 0 && !orc_is_terminal(c1, c2)) {
            double log_term = log((double)(visits[node] + 1));          /* mcts.py:29 */
            double best_score = -INFINITY;
            int best = -1;
            for (int ch = first[node]; ch < first[node] + nchild[node]; ++ch) {
                double d = (double)(visits[ch] + 1);
                double score = (value[ch] / d) + c_puct * sqrt(log_term / d);
                if (score > best_score) { best_score = score; best = ch; }
            }
            node = best;
            orc_drop(&c1, &c2, player, action[node]);   /* always the root mover's colour */
            ++depth_sum;
        }
        double reward;
        if (orc_is_terminal(c1, c2)) {
            reward = orc_reward(c1, c2);
            ++term;
        } else {
            int legal = orc_legal_mask(c1, c2);
            /* planes from agent.current_player (`side`), not from the mover (agent_code.py:85-87) */
            u64 me = side == 1 ? c1 : c2, opp = side == 1 ? c2 : c1;
            if (eval_kind == ORC_EVAL_ROLLOUT) {
                /* no counterpart in the reference: a uniformly random playout whose value is the
                   final position's get_reward() (absolute frame), the root mover moving first */
                int res = orc_rollout(c1, c2, player, seed, game_id, move_index, (uint32_t)s);
                reward = res == 1 ? -1.0 : (res == -1 ? 1.0 : 0.0);
            } else {
                reward = eval_leaf(eval_kind, fn, ctx, me, opp);
            }
            ++evals;
            first[node] = n_nodes;
            for (int c = 0; c < ORC_COLS; ++c) {
                if (!((legal >> c) & 1)) continue;
                visits[n_nodes] = 0; value[n_nodes] = 0.0; parent[n_nodes] = node;
                first[n_nodes] = -1; nchild[n_nodes] = 0; action[n_nodes] = (int8_t)c;
                ++n_nodes; ++nchild[node];
            }
        }
        for (int n = node; n >= 0; n = parent[n]) { visits[n] += 1; value[n] += reward; }
    }

    int best_a = -1, best_v = -1;
    for (int c = 0; c < ORC_COLS; ++c) visits_out[c] = 0;
    for (int ch = first[0]; nchild[0] > 0 && ch < first[0] + nchild[0]; ++ch) {
        visits_out[action[ch]] = visits[ch];
        if (visits[ch] > best_v) { best_v = visits[ch]; best_a = action[ch]; }
    }
    if (st) {
        st->nodes = n_nodes; st->depth_sum = depth_sum; st->evals = evals; st->terminal_hits = term;
        st->root_visits = visits[0]; st->root_value = value[0];
    }
    free(visits); free(value); free(parent); free(first); free(nchild); free(action);
    return best_a;
}

int orc_compat_search(u64 b1, u64 b2, int player, int sims, double c_puct, int eval_kind,
                      orc_eval_fn fn, void *ctx, int32_t *visits_out, orc_stats *st) {
    return compat_search_impl(b1, b2, player, player, sims, c_puct, eval_kind, fn, ctx, 0, 0, 0, visits_out, st);
}

/* the same search with the agent's side (agent.current_player) given apart from the mover */
int orc_compat_search_sided(u64 b1, u64 b2, int player, int side, int sims, double c_puct, int eval_kind,
                            orc_eval_fn fn, void *ctx, int32_t *visits_out, orc_stats *st) {
    return compat_search_impl(b1, b2, player, side, sims, c_puct, eval_kind, fn, ctx, 0, 0, 0, visits_out, st);
}

/* same search with the rollout evaluator's RNG coordinates (seed, game, move index) */
int orc_compat_search_seeded(u64 b1, u64 b2, int player, int sims, double c_puct, int eval_kind, u64 seed,
                             uint32_t game_id, uint32_t move_index, int32_t *visits_out, orc_stats *st) {
    return compat_search_impl(b1, b2, player, player, sims, c_puct, eval_kind, NULL, NULL, seed, game_id, move_index,
                              visits_out, st);
}

/*
 * train_alpha_zero.py:36-48 - one self-play game from (b1,b2,player) with the
 * same agent on both sides.  moves_out/visits_out hold up to 42 plies.
 * Returns the number of plies played; *result_out gets get_result() (2 = None
 * when max_plies stopped the game early).
 */
int orc_selfplay_game(u64 b1, u64 b2, int player, int sims, double c_puct, int eval_kind,
                      orc_eval_fn fn, void *ctx, int max_plies, int8_t *moves_out,
                      int32_t *visits_out, int *result_out, u64 *final_out, orc_stats *total) {
    int ply = 0;
    orc_stats acc; memset(&acc, 0, sizeof acc);
    while (!orc_is_terminal(b1, b2) && ply < max_plies) {
        orc_stats st;
        int a = orc_compat_search(b1, b2, player, sims, c_puct, eval_kind, fn, ctx,
                                  visits_out + 7 * ply, &st);
        acc.nodes += st.nodes; acc.depth_sum += st.depth_sum; acc.evals += st.evals;
        acc.terminal_hits += st.terminal_hits;
        if (a < 0) break;
        moves_out[ply++] = (int8_t)a;
        orc_drop(&b1, &b2, player, a);
        player = 3 - player;
    }
    *result_out = orc_result(b1, b2);
    final_out[0] = b1; final_out[1] = b2;
    if (total) *total = acc;
    return ply;
}

/* =======================================================================
 * PUCT search ("AZ" mode) and the random-rollout evaluator.
 *
 * The reference has neither (its search ignores priors, mcts.py:35-38, and has
 * no rollouts), so nothing below is pinned by the reference: PARITY UNPINNED.
 * This is the specification the device kernels are held to, bit for bit:
 *   - node statistics N (visits) and W (value sum seen by the player who moved
 *     INTO the node); children contiguous, ascending column;
 *   - selection at a node with N_p visits: argmax over children of
 *        Q + U,  Q = N>0 ? W/N : 0,  U = c_puct * P * sqrt(N_p) / (1 + N)
 *     in float64 in exactly this operation order, lowest column on ties;
 *   - descent alternates colours; a leaf is evaluated from the side to move
 *     (value v in [-1,1], priors over legal columns); a terminal leaf has
 *     v = -1 if the previous mover completed four, else 0;
 *   - backup: leaf gets W += -v, its parent W += +v, alternating to the root,
 *     every node on the path N += 1;
 *   - root priors are mixed with caller-supplied noise when the root is
 *     expanded: P = (float)((1-eps)*P + eps*noise);
 *   - move = most visited root child, lowest column on ties.
 * ===================================================================== */

typedef void (*orc_az_eval_fn)(void *ctx, u64 me, u64 opp, int legal, float *priors7, float *value);

void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Uniformly random playout from (b1,b2) with `mover` to move, one Philox word per ply:
 * counter = (game_id, move_index, sim, rollout_ply/4), key = seed.  Returns orc_result(). */
int orc_rollout(u64 b1, u64 b2, int mover, u64 seed, uint32_t game_id, uint32_t move_index, uint32_t sim) {
    uint32_t rnd[4];
    for (int rp = 0; rp < 42; ++rp) {
        if (orc_is_terminal(b1, b2)) break;
        int legal = orc_legal_mask(b1, b2);
        if (!legal) break;
        if ((rp & 3) == 0)
            orc_philox4x32(game_id, move_index, sim, (uint32_t)(rp >> 2), (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
        int nl = __builtin_popcount(legal);
        int idx = (int)(((uint64_t)rnd[rp & 3] * (uint64_t)nl) >> 32);
        int m = legal;
        for (int k = 0; k < idx; ++k) m &= m - 1;
        orc_drop(&b1, &b2, mover, __builtin_ctz(m));
        mover = 3 - mover;
    }
    return orc_result(b1, b2);
}

/* priors of the fixed evaluators: uniform, or weights 1 + byte_a of the position hash */
void orc_fixed_priors(int kind, u64 me, u64 opp, int legal, float *p) {
    int nl = __builtin_popcount(legal);
    if (kind == ORC_EVAL_HASH_Q8 || kind == ORC_EVAL_HASH_F32) {
        u64 h = orc_mix64(me, opp);
        float sum = 0.0f;
        for (int a = 0; a < 7; ++a) {
            p[a] = ((legal >> a) & 1) ? (float)(1 + (int)((h >> (8 * a)) & 0xFF)) : 0.0f;
            sum += p[a];
        }
        for (int a = 0; a < 7; ++a) p[a] = p[a] / sum;
    } else {
        for (int a = 0; a < 7; ++a) p[a] = ((legal >> a) & 1) ? (float)(1.0 / (double)nl) : 0.0f;
    }
}

int orc_az_search(u64 b1, u64 b2, int player, int sims, double c_puct, int eval_kind, orc_az_eval_fn fn,
                  void *ctx, const float *root_noise, double eps, u64 seed, uint32_t game_id,
                  uint32_t move_index, int32_t *visits_out, orc_stats *st) {
    int cap = 1 + 7 * sims;
    int32_t *visits = (int32_t *)calloc(cap, sizeof(int32_t));
    double *value = (double *)calloc(cap, sizeof(double));
    float *prior = (float *)calloc(cap, sizeof(float));
    int32_t *parent = (int32_t *)malloc(cap * sizeof(int32_t));
    int32_t *first = (int32_t *)malloc(cap * sizeof(int32_t));
    int8_t *nchild = (int8_t *)calloc(cap, 1);
    int8_t *action = (int8_t *)malloc(cap);
    int n_nodes = 1;
    int64_t depth_sum = 0, evals = 0, term = 0;
    parent[0] = -1; first[0] = -1; action[0] = -1;

    for (int s = 0; s < sims; ++s) {
        int node = 0, mover = player, depth = 0;
        u64 c1 = b1, c2 = b2;
        while (nchild[node] > 0 && !orc_is_terminal(c1, c2)) {
            double sq = sqrt((double)visits[node]);
            double best_score = -INFINITY;
            int best = -1;
            for (int ch = first[node]; ch < first[node] + nchild[node]; ++ch) {
                double n = (double)visits[ch];
                double q = visits[ch] > 0 ? value[ch] / n : 0.0;
                double u = c_puct * (double)prior[ch] * sq / (1.0 + n);
                double score = q + u;
                if (score > best_score) { best_score = score; best = ch; }
            }
            node = best;
            orc_drop(&c1, &c2, mover, action[node]);
            mover = 3 - mover;
            ++depth;
        }
        depth_sum += depth;
        double v;                                   /* value for the side to move at the leaf */
        int legal = orc_legal_mask(c1, c2);
        if (orc_is_terminal(c1, c2) || !legal) {
            v = (orc_has_four(c1) || orc_has_four(c2)) ? -1.0 : 0.0;
            ++term;
        } else {
            u64 me = mover == 1 ? c1 : c2, opp = mover == 1 ? c2 : c1;
            float p[7], fv = 0.0f;
            if (eval_kind == ORC_EVAL_CALLBACK) {
                fn(ctx, me, opp, legal, p, &fv);
                v = (double)fv;
            } else if (eval_kind == ORC_EVAL_ROLLOUT) {
                orc_fixed_priors(ORC_EVAL_CONST, me, opp, legal, p);
                int res = orc_rollout(c1, c2, mover, seed, game_id, move_index, (uint32_t)s);
                v = (res == 2 || res == 0) ? 0.0 : ((res == 1) == (mover == 1) ? 1.0 : -1.0);
            } else {
                orc_fixed_priors(eval_kind, me, opp, legal, p);
                v = eval_leaf(eval_kind, NULL, NULL, me, opp);
            }
            ++evals;
            first[node] = n_nodes;
            for (int c = 0; c < ORC_COLS; ++c) {
                if (!((legal >> c) & 1)) continue;
                float pc = p[c];
                if (node == 0 && root_noise)
                    pc = (float)((1.0 - eps) * (double)pc + eps * (double)root_noise[c]);
                visits[n_nodes] = 0; value[n_nodes] = 0.0; prior[n_nodes] = pc; parent[n_nodes] = node;
                first[n_nodes] = -1; nchild[n_nodes] = 0; action[n_nodes] = (int8_t)c;
                ++n_nodes; ++nchild[node];
            }
        }
        double sign = -1.0;
        for (int n = node; n >= 0; n = parent[n]) { visits[n] += 1; value[n] += sign * v; sign = -sign; }
    }

    int best_a = -1, best_v = -1;
    for (int c = 0; c < ORC_COLS; ++c) visits_out[c] = 0;
    for (int ch = first[0]; nchild[0] > 0 && ch < first[0] + nchild[0]; ++ch) {
        visits_out[action[ch]] = visits[ch];
        if (visits[ch] > best_v) { best_v = visits[ch]; best_a = action[ch]; }
    }
    if (st) {
        st->nodes = n_nodes; st->depth_sum = depth_sum; st->evals = evals; st->terminal_hits = term;
        st->root_visits = visits[0]; st->root_value = value[0];
    }
    free(visits); free(value); free(prior); free(parent); free(first); free(nchild); free(action);
    return best_a;
}


/* =======================================================================
 * Depth-limited minimax opponent (nnbattle/agents/minimax/agent_code.py:34-89) with the
 * position score of connect_four_game.py:215-278, on bitboards.
 *
 * The reference agent always maximises for AI_PIECE (= 2), whoever is to move; here the
 * maximising side is the root mover, which is the same thing whenever colour 2 is to move
 * (the case the golden vectors cover).  Win / loss are +-ORC_MM_INF (the reference's
 * +-math.inf).  When every move loses inside the horizon the reference returns
 * random.choice(valid_locations); this restatement returns the lowest legal column.
 * ===================================================================== */
#define ORC_MM_INF (1 << 30)

static int mm_bit(int row_from_top, int col) { return col * 7 + (5 - row_from_top); }

/* the 69 windows in the order of connect_four_game.py:228-254 (the order does not matter for a sum) */
static int mm_windows(u64 *out) {
    int n = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 4; ++c) { u64 m = 0; for (int i = 0; i < 4; ++i) m |= 1ULL << mm_bit(r, c + i); out[n++] = m; }
    for (int c = 0; c < 7; ++c)
        for (int r = 0; r < 3; ++r) { u64 m = 0; for (int i = 0; i < 4; ++i) m |= 1ULL << mm_bit(r + i, c); out[n++] = m; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) { u64 m = 0; for (int i = 0; i < 4; ++i) m |= 1ULL << mm_bit(r + i, c + i); out[n++] = m; }
    for (int r = 3; r < 6; ++r)
        for (int c = 0; c < 4; ++c) { u64 m = 0; for (int i = 0; i < 4; ++i) m |= 1ULL << mm_bit(r - i, c + i); out[n++] = m; }
    return n;
}

/* score_position(piece) with evaluate_window (connect_four_game.py:256-278) */
int orc_score_position(u64 b1, u64 b2, int piece) {
    static u64 win[69];
    static int nwin = 0;
    if (!nwin) nwin = mm_windows(win);
    const u64 me = piece == 1 ? b1 : b2, opp = piece == 1 ? b2 : b1, empty = ~(b1 | b2);
    int score = 3 * __builtin_popcountll(me & (0x3FULL << (3 * 7)));      /* centre column */
    for (int i = 0; i < nwin; ++i) {
        const int a = __builtin_popcountll(me & win[i]), o = __builtin_popcountll(opp & win[i]);
        const int e = __builtin_popcountll(empty & win[i]);
        if (a == 4) score += 100;
        else if (a == 3 && e == 1) score += 5;
        else if (a == 2 && e == 2) score += 2;
        if (o == 3 && e == 1) score -= 4;
    }
    return score;
}

static int mm_search(u64 b1, u64 b2, int mover, int me, int depth, int alpha, int beta, int maximizing, int *col_out) {
    const u64 mine = me == 1 ? b1 : b2, theirs = me == 1 ? b2 : b1;
    *col_out = -1;
    if (orc_has_four(mine)) return ORC_MM_INF;
    if (orc_has_four(theirs)) return -ORC_MM_INF;
    const int legal = orc_legal_mask(b1, b2);
    if (!legal) return 0;
    if (depth == 0) return orc_score_position(b1, b2, me);
    int value = maximizing ? -ORC_MM_INF - 1 : ORC_MM_INF + 1;
    for (int c = 0; c < ORC_COLS; ++c) {
        if (!((legal >> c) & 1)) continue;
        u64 c1 = b1, c2 = b2;
        orc_drop(&c1, &c2, mover, c);
        int sub;
        const int s = mm_search(c1, c2, 3 - mover, me, depth - 1, alpha, beta, !maximizing, &sub);
        if (maximizing) {
            if (s > value) { value = s; *col_out = c; }
            if (value > alpha) alpha = value;
        } else {
            if (s < value) { value = s; *col_out = c; }
            if (value < beta) beta = value;
        }
        if (alpha >= beta) break;
    }
    return value;
}

/* MinimaxAgent.select_move for the side to move; returns the column (-1 without a legal move or
 * in a finished position) and the root value. */
int orc_minimax(u64 b1, u64 b2, int mover, int depth, int *score_out) {
    int col;
    const int s = mm_search(b1, b2, mover, mover, depth, -ORC_MM_INF - 2, ORC_MM_INF + 2, 1, &col);
    if (score_out) *score_out = s;
    return col;
}

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

enum { ORC_EVAL_CONST = 0, ORC_EVAL_HASH_Q8 = 1, ORC_EVAL_HASH_F32 = 2, ORC_EVAL_CALLBACK = 3 };

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
 * encoded from the root mover's side (:85); absolute +-1 terminal reward
 * (:182-186) with no negation in the backup (mcts.py:40-44); priors discarded
 * (mcts.py:35-38); lowest column wins ties (mcts.py:30, agent_code.py:223).
 * Returns the selected column, or -1 when the root never got children.
 */
int orc_compat_search(u64 b1, u64 b2, int player, int sims, double c_puct, int eval_kind,
                      orc_eval_fn fn, void *ctx, int32_t *visits_out, orc_stats *st) {
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
            u64 me = player == 1 ? c1 : c2, opp = player == 1 ? c2 : c1;
            reward = eval_leaf(eval_kind, fn, ctx, me, opp);
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

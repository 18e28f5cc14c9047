"""ctypes binding for oracle/liboracle.so (the C restatement in c4_oracle.c).

TEST INFRASTRUCTURE ONLY - see the header of c4_oracle.c.  Importable only from
tests/, the smoke check and the benchmark's CPU-baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

EVAL_CONST, EVAL_HASH_Q8, EVAL_HASH_F32, EVAL_CALLBACK, EVAL_ROLLOUT = 0, 1, 2, 3, 4
EVAL_KINDS = {"const": EVAL_CONST, "hash_q8": EVAL_HASH_Q8, "hash_f32": EVAL_HASH_F32,
              "rollout": EVAL_ROLLOUT}

EVAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_uint64, C.c_uint64)
AZ_EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_float),
                         C.POINTER(C.c_float))


class Stats(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("depth_sum", C.c_int64), ("evals", C.c_int64),
                ("terminal_hits", C.c_int64), ("root_visits", C.c_int32), ("root_value", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c4_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64 = C.c_uint64
        L.orc_legal_mask.argtypes = [u64, u64]
        L.orc_drop.argtypes = [C.POINTER(u64), C.POINTER(u64), C.c_int, C.c_int]
        L.orc_has_four.argtypes = [u64]
        L.orc_is_full.argtypes = [u64, u64]
        L.orc_is_terminal.argtypes = [u64, u64]
        L.orc_reward.argtypes = [u64, u64]
        L.orc_reward.restype = C.c_double
        L.orc_winner.argtypes = [u64, u64]
        L.orc_result.argtypes = [u64, u64]
        L.orc_mix64.argtypes = [u64, u64]
        L.orc_mix64.restype = u64
        L.orc_hash_value_q8.argtypes = [u64, u64]
        L.orc_hash_value_q8.restype = C.c_double
        L.orc_hash_value_f32.argtypes = [u64, u64]
        L.orc_hash_value_f32.restype = C.c_double
        L.orc_compat_search.argtypes = [u64, u64, C.c_int, C.c_int, C.c_double, C.c_int, EVAL_FN,
                                        C.c_void_p, C.POINTER(C.c_int32), C.POINTER(Stats)]
        L.orc_compat_search_sided.argtypes = [u64, u64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, EVAL_FN,
                                              C.c_void_p, C.POINTER(C.c_int32), C.POINTER(Stats)]
        L.orc_selfplay_game.argtypes = [u64, u64, C.c_int, C.c_int, C.c_double, C.c_int, EVAL_FN,
                                        C.c_void_p, C.c_int, C.POINTER(C.c_int8),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(u64),
                                        C.POINTER(Stats)]
        L.orc_compat_search_seeded.argtypes = [u64, u64, C.c_int, C.c_int, C.c_double, C.c_int, u64, C.c_uint32,
                                               C.c_uint32, C.POINTER(C.c_int32), C.POINTER(Stats)]
        L.orc_az_search.argtypes = [u64, u64, C.c_int, C.c_int, C.c_double, C.c_int, AZ_EVAL_FN, C.c_void_p,
                                    C.POINTER(C.c_float), C.c_double, u64, C.c_uint32, C.c_uint32,
                                    C.POINTER(C.c_int32), C.POINTER(Stats)]
        L.orc_rollout.argtypes = [u64, u64, C.c_int, u64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_philox4x32.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.orc_fixed_priors.argtypes = [C.c_int, u64, u64, C.c_int, C.POINTER(C.c_float)]
        L.orc_score_position.argtypes = [u64, u64, C.c_int]
        L.orc_minimax.argtypes = [u64, u64, C.c_int, C.c_int, C.POINTER(C.c_int)]
        _lib = L
    return _lib


_NULL_FN = EVAL_FN(0)


def _eval_args(evaluator):
    """evaluator: a name from EVAL_KINDS or a Python callable(me_bits, opp_bits) -> float."""
    if isinstance(evaluator, str):
        return EVAL_KINDS[evaluator], _NULL_FN, None
    cb = EVAL_FN(lambda ctx, me, opp: float(evaluator(me, opp)))
    return EVAL_CALLBACK, cb, cb       # third item keeps the callback alive


def legal_mask(b1, b2):
    return lib().orc_legal_mask(b1, b2)


def drop(b1, b2, player, col):
    x1, x2 = C.c_uint64(b1), C.c_uint64(b2)
    ok = lib().orc_drop(C.byref(x1), C.byref(x2), player, col)
    return bool(ok), x1.value, x2.value


def compat_search(b1, b2, player, sims, c_puct, evaluator, side=None):
    """side: the colour the evaluator looks from (agent.current_player); None = the mover."""
    kind, fn, _keep = _eval_args(evaluator)
    visits = (C.c_int32 * 7)()
    st = Stats()
    a = lib().orc_compat_search_sided(b1, b2, player, player if side is None else side, sims, c_puct, kind, fn,
                                      None, visits, C.byref(st))
    return (None if a < 0 else a), list(visits), st


def selfplay_game(sims, c_puct, evaluator, b1=0, b2=0, player=1, max_plies=42):
    kind, fn, _keep = _eval_args(evaluator)
    moves = (C.c_int8 * 42)()
    visits = (C.c_int32 * (42 * 7))()
    result = C.c_int()
    final = (C.c_uint64 * 2)()
    st = Stats()
    n = lib().orc_selfplay_game(b1, b2, player, sims, c_puct, kind, fn, None, max_plies, moves,
                                visits, C.byref(result), final, C.byref(st))
    v = np.array(visits, dtype=np.int32).reshape(42, 7)[:n]
    res = None if result.value == 2 else result.value
    return [int(m) for m in moves[:n]], v.tolist(), res, (final[0], final[1]), st


def compat_search_seeded(b1, b2, player, sims, c_puct, evaluator, seed=0, game_id=0, move_index=0):
    """compat search whose rollout evaluator draws at RNG coordinates (seed, game_id, move_index)."""
    visits = (C.c_int32 * 7)()
    st = Stats()
    a = lib().orc_compat_search_seeded(b1, b2, player, sims, c_puct, EVAL_KINDS[evaluator], seed, game_id,
                                       move_index, visits, C.byref(st))
    return (None if a < 0 else a), list(visits), st


_NULL_AZ_FN = AZ_EVAL_FN(0)


def az_search(b1, b2, player, sims, c_puct, evaluator, root_noise=None, eps=0.0, seed=0, game_id=0,
              move_index=0):
    """PUCT search.  evaluator: a name from EVAL_KINDS, or callable(me, opp, legal) -> (priors[7], value)."""
    visits = (C.c_int32 * 7)()
    st = Stats()
    noise = None
    if root_noise is not None:
        noise = (C.c_float * 7)(*[float(x) for x in root_noise])
    if isinstance(evaluator, str):
        kind, fn = EVAL_KINDS[evaluator], _NULL_AZ_FN
    else:
        def _cb(ctx, me, opp, legal, pri, val):
            p, v = evaluator(me, opp, legal)
            for i in range(7):
                pri[i] = float(p[i])
            val[0] = float(v)
        kind, fn = EVAL_CALLBACK, AZ_EVAL_FN(_cb)
    a = lib().orc_az_search(b1, b2, player, sims, c_puct, kind, fn, None, noise, eps, seed, game_id, move_index,
                            visits, C.byref(st))
    return (None if a < 0 else a), list(visits), st


MM_INF = 1 << 30


def score_position(b1, b2, piece):
    return lib().orc_score_position(b1, b2, piece)


def minimax(b1, b2, mover, depth):
    """(column or None, root value) of the depth-limited minimax opponent for the side to move."""
    score = C.c_int()
    col = lib().orc_minimax(b1, b2, mover, depth, C.byref(score))
    return (None if col < 0 else col), score.value

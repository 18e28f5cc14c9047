"""ctypes binding of libnnb.so (the C ABI declared in include/nnb.h).

There is no CPU fallback: if the library is missing, importing the engine fails
with instructions to build it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NNB_LIB selects another build of the same ABI (tools/build_variants.py: timing experiments)
LIB_PATH = os.environ.get("NNB_LIB") or os.path.join(HERE, "libnnb.so")

MODE_COMPAT, MODE_AZ = 0, 1
EVAL_NET, EVAL_CONST, EVAL_HASH_Q8, EVAL_HASH_F32, EVAL_ROLLOUT = 0, 1, 2, 3, 4
EVALUATORS = {"net": EVAL_NET, "const": EVAL_CONST, "hash_q8": EVAL_HASH_Q8,
              "hash_f32": EVAL_HASH_F32, "rollout": EVAL_ROLLOUT}
FLAG_LOCKSTEP = 1
NET_F16, NET_F32, NET_F32S = 0, 1, 2
PRECISIONS = {"f16": NET_F16, "f32": NET_F32, "f32s": NET_F32S}
MODES = {"compat": MODE_COMPAT, "az": MODE_AZ}


class Config(C.Structure):
    _fields_ = [("max_games", C.c_int32), ("max_sims", C.c_int32), ("device", C.c_int32),
                ("n_weight_sets", C.c_int32), ("net_precision", C.c_int32), ("eval_cache_slots", C.c_int32)]


class SearchParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("evaluator", C.c_int32), ("sims", C.c_int32),
                ("weight_set", C.c_int32), ("flags", C.c_int32), ("rng_move_index", C.c_int32),
                ("rng_game_base", C.c_int32), ("weight_set_colour2_plus1", C.c_int32),
                ("c_puct", C.c_double), ("dirichlet_alpha", C.c_double),
                ("dirichlet_eps", C.c_double), ("seed", C.c_uint64),
                ("side_of_colour", C.c_int32 * 2), ("round_limit", C.c_int64)]


class RoundStats(C.Structure):
    _fields_ = [("games", C.c_int64), ("plies", C.c_int64), ("sims", C.c_int64),
                ("net_leaves", C.c_int64), ("terminal_hits", C.c_int64), ("depth_sum", C.c_int64),
                ("steps", C.c_int64), ("moves_rounds", C.c_int64), ("cache_hits", C.c_int64),
                ("chains", C.c_int64), ("finished_total", C.c_int64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


TRAJECTORY_BYTES = 32 + 42 * 32
PLY_BYTES = 32

# name -> (restype, argtypes); must list every function include/nnb.h declares
_P = C.c_void_p
SIGNATURES = {
    "nnb_version": (C.c_char_p, []),
    "nnb_last_error": (C.c_char_p, []),
    "nnb_workspace_bytes": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_size_t)]),
    "nnb_engine_create": (C.c_int, [C.POINTER(Config), _P, C.c_size_t, _P, C.POINTER(_P)]),
    "nnb_engine_destroy": (C.c_int, [_P]),
    "nnb_load_weights": (C.c_int, [_P, C.c_int, _P] + [_P] * 12),
    "nnb_game_ops": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "nnb_encode_planes": (C.c_int, [_P, C.c_int32, _P, _P]),
    "nnb_minimax": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P]),
    "nnb_ucb_scores": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_double, _P]),
    "nnb_net_forward": (C.c_int, [_P, C.c_int, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "nnb_net_forward_planes": (C.c_int, [_P, C.c_int, _P, C.c_int32, _P, _P, _P]),
    "nnb_debug_activations": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int32, _P]),
    "nnb_search": (C.c_int, [_P, _P, C.POINTER(SearchParams), C.c_int32, _P, _P, _P, _P, _P, _P, _P,
                             C.POINTER(RoundStats)]),
    "nnb_search_static": (C.c_int, [_P, C.POINTER(SearchParams), C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "nnb_dirichlet_noise": (C.c_int, [_P, C.c_int32, C.c_uint64, _P, _P, _P, C.c_double, _P]),
    "nnb_selfplay_round": (C.c_int, [_P, _P, C.POINTER(SearchParams), C.c_int32, C.c_int32, _P, _P,
                                     _P, C.POINTER(RoundStats)]),
    "nnb_selfplay_stream": (C.c_int, [_P, _P, C.POINTER(SearchParams), C.c_int32, C.c_int32, C.c_int32, _P, _P,
                                      C.c_int32, _P, _P, C.c_int32, C.POINTER(RoundStats)]),
    "nnb_trajectory_samples": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "nnb_profile_enable": (C.c_int, [_P, C.c_int]),
    "nnb_profile_reset": (C.c_int, [_P]),
    "nnb_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nnb_profile_clocks": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "nnb_debug_second_stream": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
}

_lib = None


class NnbError(RuntimeError):
    pass


def load():
    """Loads libnnb.so and binds every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own copy of the HIP runtime.  Whichever copy is mapped first serves the whole process, and a
    # second, separately initialised one finds no device: torch's has to be the one, since the engine runs on torch's
    # streams and memory.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NnbError(
            f"{LIB_PATH} is missing: build it with `python -m alphazero_amd.build` "
            "(hipcc, --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().nnb_last_error().decode(errors="replace")
        raise NnbError(f"{what} failed ({rc}): {msg}")

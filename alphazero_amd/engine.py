"""Python handle on the device engine in libnnb.so.

PyTorch is used for device memory and streams only: every array handed to the
library is a torch tensor on the engine's device, and every kernel runs on torch's
current stream for that device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import (EVALUATORS, MODES, PRECISIONS, TRAJECTORY_BYTES, Config, NnbError, RoundStats,
                   SearchParams, check)

STATE_DICT_KEYS = [
    "conv_layers.0.weight", "conv_layers.0.bias",
    "conv_layers.2.weight", "conv_layers.2.bias",
    "conv_layers.4.weight", "conv_layers.4.bias",
    "fc_layers.0.weight", "fc_layers.0.bias",
    "policy_head.weight", "policy_head.bias",
    "value_head.weight", "value_head.bias",
]
STATE_DICT_SHAPES = [(128, 2, 4, 4), (128,), (128, 128, 4, 4), (128,), (128, 128, 4, 4), (128,),
                     (1024, 11520), (1024,), (7, 1024), (7,), (1, 1024), (1,)]

PROFILE_CLASSES = ["tree_step", "conv0", "conv2", "conv4", "fc", "heads", "readout"]

# numpy view of one nnb_trajectory record (include/nnb.h)
PLY_DTYPE = np.dtype([("colour_boards", "<u8", (2,)), ("visits", "<u2", (7,)), ("action", "u1"),
                      ("player", "u1")])
TRAJECTORY_DTYPE = np.dtype([("length", "<i4"), ("result", "<i4"), ("game_id", "<i4"),
                             ("reserved", "<i4"), ("final_boards", "<u8", (2,)),
                             ("plies", PLY_DTYPE, (42,))])
assert PLY_DTYPE.itemsize == 32 and TRAJECTORY_DTYPE.itemsize == TRAJECTORY_BYTES


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One engine = one GPU, one workspace, up to `max_games` concurrent games."""

    def __init__(self, max_games: int, max_sims: int, device: Optional[torch.device] = None,
                 n_weight_sets: int = 1, precision: str = "f32s", eval_cache_slots: int = 0):
        if not torch.cuda.is_available():
            raise NnbError("no GPU visible: the engine runs on an MI355X only (there is no CPU path)")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.max_games, self.max_sims = int(max_games), int(max_sims)
        self.precision = precision
        self.cfg = Config(self.max_games, self.max_sims, self.device.index, int(n_weight_sets),
                          PRECISIONS[precision], int(eval_cache_slots))
        nbytes = C.c_size_t()
        check(self.lib.nnb_workspace_bytes(C.byref(self.cfg), C.byref(nbytes)), "nnb_workspace_bytes")
        self.workspace_bytes = nbytes.value
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(self.workspace_bytes + 256, dtype=torch.uint8, device=self.device)
            base = self.workspace.data_ptr()
            self._ws_ptr = (base + 255) // 256 * 256
            handle = C.c_void_p()
            check(self.lib.nnb_engine_create(C.byref(self.cfg), C.c_void_p(self._ws_ptr), nbytes,
                                             self._stream(), C.byref(handle)), "nnb_engine_create")
        self.handle = handle

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nnb_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, x, dtype) -> torch.Tensor:
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        arr = np.ascontiguousarray(x)
        if arr.dtype == np.uint64:                    # torch has no general uint64 support
            arr = arr.view(np.int64)
        return torch.from_numpy(arr).to(device=self.device, dtype=dtype).contiguous()

    def params(self, sims: int, c_puct: float = 1.4, evaluator="net", mode="compat", weight_set: int = 0,
               dirichlet_alpha: float = 0.0, dirichlet_eps: float = 0.0, seed: int = 0,
               lockstep: bool = False, rng_move_index: int = 0, rng_game_base: int = 0,
               weight_set_colour2: Optional[int] = None, separate_heads: bool = False,
               sync_moves: bool = False, eval_cache: bool = False, side_of_colour=(0, 0),
               round_limit: int = 0, one_stream: bool = False, two_streams: bool = False,
               separate_convs: bool = False) -> SearchParams:
        ev = EVALUATORS[evaluator] if isinstance(evaluator, str) else int(evaluator)
        md = MODES[mode] if isinstance(mode, str) else int(mode)
        return SearchParams(md, ev, int(sims), int(weight_set), int(bool(lockstep)) | (2 if separate_heads else 0) | (4 if sync_moves else 0)
                            | (8 if eval_cache else 0) | (16 if one_stream else 0) | (32 if two_streams else 0)
                            | (64 if separate_convs else 0),
                            int(rng_move_index),
                            int(rng_game_base), 0 if weight_set_colour2 is None else int(weight_set_colour2) + 1,
                            float(c_puct), float(dirichlet_alpha), float(dirichlet_eps),
                            int(seed), (C.c_int32 * 2)(int(side_of_colour[0]), int(side_of_colour[1])),
                            int(round_limit))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, weight_set: int = 0):
        """Connect4Net state dict (nnbattle/agents/alphazero/network.py:28-41) -> device layouts."""
        tensors = []
        for key, shape in zip(STATE_DICT_KEYS, STATE_DICT_SHAPES):
            if key not in state_dict:
                raise KeyError(f"state dict lacks {key}")
            t = state_dict[key].detach().to(device=self.device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{key}: shape {tuple(t.shape)}, expected {shape}")
            tensors.append(t)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_load_weights(self.handle, weight_set, self._stream(),
                                            *[_ptr(t) for t in tensors]), "nnb_load_weights")

    # ------------------------------------------------------------------ K1 / K3
    def game_ops(self, colour_boards, players=None, actions=None):
        """Returns (boards_out uint64[n,2], legal uint8[n], status uint8[n]) as numpy arrays."""
        b = self._dev(np.asarray(colour_boards, dtype=np.uint64).reshape(-1, 2), torch.int64)
        n = b.shape[0]
        pl = self._dev(np.asarray(players, dtype=np.int8), torch.int8) if players is not None else None
        ac = self._dev(np.asarray(actions, dtype=np.int8), torch.int8) if actions is not None else None
        out = torch.empty_like(b)
        legal = torch.empty(n, dtype=torch.uint8, device=self.device)
        status = torch.empty(n, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_game_ops(self._stream(), n, _ptr(b), _ptr(pl), _ptr(ac), _ptr(out),
                                        _ptr(legal), _ptr(status)), "nnb_game_ops")
        return out.cpu().numpy().view(np.uint64), legal.cpu().numpy(), status.cpu().numpy()

    def minimax(self, colour_boards, movers, depth: int):
        """Depth-limited minimax move for the side to move of every position.
        Returns (column int8[n] (-1 = none), root value int32[n]) as numpy arrays."""
        b = self._dev(np.asarray(colour_boards, dtype=np.uint64).reshape(-1, 2), torch.int64)
        n = b.shape[0]
        mv = self._dev(np.asarray(movers, dtype=np.int8), torch.int8)
        col = torch.empty(n, dtype=torch.int8, device=self.device)
        score = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_minimax(self._stream(), n, _ptr(b), _ptr(mv), int(depth), _ptr(col), _ptr(score)),
                  "nnb_minimax")
        return col.cpu().numpy(), score.cpu().numpy()

    def encode_planes(self, side_boards) -> torch.Tensor:
        b = self._dev(np.asarray(side_boards, dtype=np.uint64).reshape(-1, 2), torch.int64)
        n = b.shape[0]
        planes = torch.empty((n, 2, 6, 7), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_encode_planes(self._stream(), n, _ptr(b), _ptr(planes)), "nnb_encode_planes")
        return planes

    def ucb_scores(self, value_sum, visits, parent_visits, c_puct: float) -> np.ndarray:
        v = self._dev(np.asarray(value_sum, dtype=np.float64), torch.float64)
        n = v.shape[0]
        vi = self._dev(np.asarray(visits, dtype=np.int32), torch.int32)
        pv = self._dev(np.asarray(parent_visits, dtype=np.int32), torch.int32)
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_ucb_scores(self.handle, self._stream(), n, _ptr(v), _ptr(vi), _ptr(pv),
                                          float(c_puct), _ptr(out)), "nnb_ucb_scores")
        return out.cpu().numpy()

    # ------------------------------------------------------------------ K4-K6
    def net_forward(self, side_boards, legal_masks=None, weight_set: int = 0):
        """side boards uint64[n,2] -> (log_policy [n,7], value [n], priors [n,7] or None), device tensors."""
        if isinstance(side_boards, torch.Tensor):
            b = side_boards.to(self.device).contiguous()
        else:
            b = self._dev(np.asarray(side_boards, dtype=np.uint64).reshape(-1, 2), torch.int64)
        n = b.shape[0]
        logp = torch.empty((n, 7), dtype=torch.float32, device=self.device)
        value = torch.empty((n,), dtype=torch.float32, device=self.device)
        legal = priors = None
        if legal_masks is not None:
            legal = self._dev(np.asarray(legal_masks, dtype=np.uint8), torch.uint8)
            priors = torch.empty((n, 7), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_net_forward(self.handle, weight_set, self._stream(), n, _ptr(b), _ptr(legal),
                                           _ptr(logp), _ptr(value), _ptr(priors)), "nnb_net_forward")
        return logp, value, priors

    def net_forward_planes(self, planes: torch.Tensor, weight_set: int = 0):
        """float planes [n,2,6,7] -> (log_policy [n,7], value [n,1]) like Connect4Net.forward."""
        x = planes.to(device=self.device, dtype=torch.float32).contiguous()
        n = x.shape[0]
        logp = torch.empty((n, 7), dtype=torch.float32, device=self.device)
        value = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_net_forward_planes(self.handle, weight_set, self._stream(), n, _ptr(x),
                                                  _ptr(logp), _ptr(value)), "nnb_net_forward_planes")
        return logp, value

    def activations(self, layer: int, n: int, weight_set: int = 0) -> torch.Tensor:
        """Hidden activations of the last net_forward call as float32: layer 2 -> [n,8,9,128],
        3 -> [n,9,10,128] (row, column, channel), 4 -> [n,1024].  Test hook."""
        shape = {2: (n, 8, 9, 128), 3: (n, 9, 10, 128), 4: (n, 1024)}[layer]
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_debug_activations(self.handle, weight_set, self._stream(), layer, n, _ptr(out)),
                  "nnb_debug_activations")
        return out

    # ------------------------------------------------------------------ K2 / K7
    def search(self, colour_boards, players, params: SearchParams, root_noise=None, side_players=None):
        """One search per root.  Returns (visits int32[n,7], action int8[n], root_value f64[n], stats).
        players: colour to move at each root (the colour the reference rule places during descent).
        side_players: colour the network looks from (agent.current_player, agent_code.py:85-87);
        None = the mover.  root_noise: optional float32 [n,7] mixed into the root priors (AZ mode)."""
        b = self._dev(np.asarray(colour_boards, dtype=np.uint64).reshape(-1, 2), torch.int64)
        n = b.shape[0]
        pl = self._dev(np.asarray(players, dtype=np.int8), torch.int8)
        visits = torch.empty((n, 7), dtype=torch.int32, device=self.device)
        action = torch.empty((n,), dtype=torch.int8, device=self.device)
        rootv = torch.empty((n,), dtype=torch.float64, device=self.device)
        stats = RoundStats()
        noise = sides = None
        if root_noise is not None:
            noise = self._dev(np.asarray(root_noise, dtype=np.float32).reshape(n, 7), torch.float32)
        if side_players is not None:
            sides = self._dev(np.asarray(side_players, dtype=np.int8).reshape(n), torch.int8)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_search(self.handle, self._stream(), C.byref(params), n, _ptr(b), _ptr(pl),
                                      _ptr(sides), _ptr(noise), _ptr(visits), _ptr(action), _ptr(rootv), C.byref(stats)),
                  "nnb_search")
        return visits.cpu().numpy(), action.cpu().numpy(), rootv.cpu().numpy(), stats.as_dict()

    def selfplay_round(self, n_games: int, params: SearchParams, first_game_id: int = 0,
                       start_boards=None, start_players=None, out: Optional[torch.Tensor] = None):
        """Plays n_games games to the end.  Returns (trajectory bytes on device [n, 1376] uint8, stats)."""
        sb = sp = None
        if start_boards is not None:
            # device tensors (int64 bit patterns of the boards, int8 movers) are used where they lie
            sb = self._dev(start_boards if isinstance(start_boards, torch.Tensor)
                           else np.asarray(start_boards, dtype=np.uint64), torch.int64).reshape(-1, 2)
            sp = self._dev(start_players if isinstance(start_players, torch.Tensor)
                           else np.asarray(start_players, dtype=np.int8), torch.int8)
            if sb.shape[0] != n_games or sp.shape[0] != n_games:
                raise ValueError("start positions must have n_games rows")
        traj = out if out is not None else torch.empty((n_games, TRAJECTORY_BYTES), dtype=torch.uint8,
                                                       device=self.device)
        stats = RoundStats()
        with torch.cuda.device(self.device):
            check(self.lib.nnb_selfplay_round(self.handle, self._stream(), C.byref(params), n_games,
                                              first_game_id, _ptr(sb), _ptr(sp), _ptr(traj),
                                              C.byref(stats)), "nnb_selfplay_round")
        return traj, stats.as_dict()

    def selfplay_stream(self, min_finished: int, params: Optional[SearchParams] = None, restart: bool = False,
                        first_game_id: int = 0, start_boards=None, start_players=None, ring_len: int = 0):
        """Self-play as an endless stream of games on all max_games slots (nnb_selfplay_stream): returns once
        `min_finished` more games have finished, with the games in flight left seated for the next call.
        restart=True seats a fresh stream (game j has id first_game_id + j and starts from start position
        j % len(start_boards), or the empty board); the ring of records it allocates holds `ring_len`
        (default 8 x max_games) records.  Returns (records of the games finished in this call, in completion
        order: device uint8 [f, 1376], a copy; stats of this call)."""
        if restart:
            if params is None:
                raise ValueError("restart needs params")
            n_ring = int(ring_len) if ring_len else 8 * self.max_games
            self._ring = torch.empty((n_ring, TRAJECTORY_BYTES), dtype=torch.uint8, device=self.device)
            self._finished_ring = torch.zeros(n_ring, dtype=torch.int32, device=self.device)
            self._stream_starts = (None, None)
            n_starts = 0
            if start_boards is not None:
                sb = self._dev(start_boards if isinstance(start_boards, torch.Tensor)
                               else np.asarray(start_boards, dtype=np.uint64), torch.int64).reshape(-1, 2)
                sp = self._dev(start_players if isinstance(start_players, torch.Tensor)
                               else np.asarray(start_players, dtype=np.int8), torch.int8)
                if sb.shape[0] != sp.shape[0] or sb.shape[0] < 1:
                    raise ValueError("start boards and players must have the same, positive number of rows")
                self._stream_starts = (sb, sp)
                n_starts = sb.shape[0]
            self._stream_n_starts = n_starts
        elif getattr(self, "_ring", None) is None:
            raise NnbError("no stream of games has been started on this engine: call with restart=True")
        sb, sp = self._stream_starts
        stats = RoundStats()
        with torch.cuda.device(self.device):
            check(self.lib.nnb_selfplay_stream(self.handle, self._stream(), C.byref(params) if params is not None else None,
                                               int(bool(restart)), int(first_game_id), int(min_finished), _ptr(sb), _ptr(sp),
                                               self._stream_n_starts, _ptr(self._ring), _ptr(self._finished_ring),
                                               self._ring.shape[0], C.byref(stats)), "nnb_selfplay_stream")
        st = stats.as_dict()
        n_ring = self._ring.shape[0]
        order = torch.arange(st["finished_total"] - st["games"], st["finished_total"], device=self.device) % n_ring
        records = self._ring[self._finished_ring[order].long()]
        return records, st

    def trajectory_samples(self, records: torch.Tensor):
        """Device records [n, 1376] uint8 -> per-ply training samples on the device (nnb_trajectory_samples):
        (planes float32 [P,2,6,7] from the mover's side, visit distributions float32 [P,7], z float32 [P]),
        P = total plies, samples ordered by record then ply."""
        rec = records.to(self.device).contiguous()
        n = rec.shape[0]
        lengths = rec[:, :4].contiguous().view(torch.int32).reshape(n).clamp(0, 42)
        offsets = (torch.cumsum(lengths, 0) - lengths).to(torch.int32)
        total = int(lengths.sum().item()) if n else 0
        planes = torch.empty((total, 2, 6, 7), dtype=torch.float32, device=self.device)
        probs = torch.empty((total, 7), dtype=torch.float32, device=self.device)
        z = torch.empty((total,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_trajectory_samples(self._stream(), n, _ptr(rec), _ptr(offsets), _ptr(planes), _ptr(probs),
                                                  _ptr(z)), "nnb_trajectory_samples")
        return planes, probs, z

    def dirichlet_noise(self, seed: int, game_ids, move_indices, legal_masks, alpha: float) -> np.ndarray:
        """The root noise the engine draws for (seed, game id, move index); float32 [n,7]."""
        gi = self._dev(np.asarray(game_ids, dtype=np.int32), torch.int32)
        mi = self._dev(np.asarray(move_indices, dtype=np.int32), torch.int32)
        lg = self._dev(np.asarray(legal_masks, dtype=np.uint8), torch.uint8)
        n = gi.shape[0]
        out = torch.empty((n, 7), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.nnb_dirichlet_noise(self._stream(), n, int(seed), _ptr(gi), _ptr(mi), _ptr(lg),
                                               float(alpha), _ptr(out)), "nnb_dirichlet_noise")
        return out.cpu().numpy()

    # ------------------------------------------------------------------ profiling
    def profile(self, on: bool = True):
        check(self.lib.nnb_profile_enable(self.handle, int(on)), "nnb_profile_enable")
        check(self.lib.nnb_profile_reset(self.handle), "nnb_profile_reset")

    def profile_clocks(self) -> dict:
        """Shader clock (MHz) of the last profiled convolution and linear-layer launch (0.0 if none was stamped)."""
        mhz = (C.c_double * 2)()
        with torch.cuda.device(self.device):
            check(self.lib.nnb_profile_clocks(self.handle, self._stream(), mhz), "nnb_profile_clocks")
        return {"conv": mhz[0], "fc": mhz[1]}

    def second_stream(self) -> tuple:
        """(runs beside the current stream, second streams discarded so far): the check every two-chain call makes
        before its first launch, run now (nnb_debug_second_stream)."""
        replaced = C.c_int()
        with torch.cuda.device(self.device):
            rc = self.lib.nnb_debug_second_stream(self.handle, self._stream(), C.byref(replaced))
        if rc < 0:
            check(rc, "nnb_debug_second_stream")
        return bool(rc), replaced.value

    def profile_report(self) -> dict:
        out = {}
        for i, name in enumerate(PROFILE_CLASSES):
            ms, cnt = C.c_double(), C.c_int64()
            check(self.lib.nnb_profile_get(self.handle, i, C.byref(ms), C.byref(cnt)), "nnb_profile_get")
            out[name] = {"avg_ms": ms.value, "samples": cnt.value}
        return out


def trajectories_to_numpy(traj: torch.Tensor) -> np.ndarray:
    """Device trajectory bytes -> structured numpy array (TRAJECTORY_DTYPE)."""
    return traj.cpu().numpy().reshape(-1).view(TRAJECTORY_DTYPE)

"""Batched self-play on the GPU engine, with the reference driver's interface.

`self_play(agent, num_games)` keeps the contract of
nnbattle/agents/alphazero/train/train_alpha_zero.py:33-55 - it returns, and appends to
`agent.memory`, one `(planes[2,6,7], zeros[7], result)` sample per game - but plays the
games concurrently: all of them advance one simulation at a time in lock step on the
device and their leaves share one forward pass.  Across GPUs the games are sharded by
index, one process per GPU, with no communication inside the search; finished
trajectories are all-gathered once per round (`gather_trajectories`).
"""
from __future__ import annotations

import logging
from typing import List, Tuple

import numpy as np
import torch

from .game import bits_to_board

logger = logging.getLogger(__name__)

MAX_CONCURRENT_GAMES = 4096      # games in flight per engine; a longer round refills slots as games end


def shard_range(num_games: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) of the games rank `rank` plays: contiguous, sizes differ by at most 1."""
    base, extra = divmod(num_games, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def gather_trajectories(local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of fixed-size trajectory records ([n, 1376] uint8, same n on every rank)
    into a [world*n, 1376] tensor on every rank.  RCCL over xGMI when `local` is a GPU
    tensor and the process group is "nccl"; also works on gloo/CPU."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def random_openings(engine, num_games: int, plies: int, seed: int = 0):
    """Start positions after `plies` uniformly random legal moves from the empty board (moves applied on
    the device with the engine's rule kernel).  The reference search is deterministic, so without this
    every self-play game from the empty board is the same game.  Returns (colour boards uint64[n,2],
    movers int8[n]); at most 6 plies, so no start position is finished."""
    if not 0 <= plies <= 6:
        raise ValueError("plies must be in [0, 6]")
    rng = np.random.RandomState(seed)
    boards = np.zeros((num_games, 2), dtype=np.uint64)
    player = 1
    for _ in range(plies):
        _, legal, _ = engine.game_ops(boards)
        pick = rng.randint(0, 1 << 30, size=num_games)
        cols = np.zeros(num_games, dtype=np.int8)
        for i in range(num_games):
            open_cols = [c for c in range(7) if (int(legal[i]) >> c) & 1]
            cols[i] = open_cols[pick[i] % len(open_cols)]
        boards, _, _ = engine.game_ops(boards, np.full(num_games, player, dtype=np.int8), cols)
        player = 3 - player
    return boards, np.full(num_games, player, dtype=np.int8)


def play_games(agent, num_games: int, first_game_id: int = 0, start_boards=None, start_players=None):
    """Plays `num_games` games on the agent's engine; returns (records, stats) where records is a
    structured numpy array (engine.TRAJECTORY_DTYPE) and stats sums the engine's round statistics."""
    from .engine import trajectories_to_numpy
    num_games = int(num_games)
    if num_games <= 0:
        return np.zeros(0, dtype=_trajectory_dtype()), {}
    eng = agent.engine(min(num_games, MAX_CONCURRENT_GAMES))
    params = eng.params(agent.num_simulations, agent.c_puct, evaluator=agent.evaluator, mode="compat",
                        eval_cache=getattr(agent, "eval_cache", False))
    traj, stats = eng.selfplay_round(num_games, params, first_game_id=first_game_id, start_boards=start_boards,
                                     start_players=start_players)
    return trajectories_to_numpy(traj), dict(stats)


def distributed_play_games(agent, num_games: int, first_game_id: int = 0, group=None, device=None):
    """One process per GPU: this rank plays its contiguous share of `num_games` (shard_range) and
    one all-gather puts every trajectory record on every rank, in game order - the replay-buffer
    exchange, the only communication of a self-play round.  Shares may differ by one game; the
    shorter ones are padded for the collective and trimmed afterwards.  Without a process group this
    is play_games.  Returns (records of all games, this rank's statistics)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return play_games(agent, num_games, first_game_id=first_game_id)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, last = shard_range(int(num_games), rank, world)
    records, stats = play_games(agent, last - first, first_game_id=first_game_id + first)
    per_rank = -(-int(num_games) // world)
    padded = np.zeros(per_rank, dtype=_trajectory_dtype())
    padded[:last - first] = records
    local = torch.from_numpy(padded.view(np.uint8).reshape(per_rank, -1))
    if device is not None:
        local = local.to(device)
    everything = gather_trajectories(local, group).cpu().numpy().reshape(-1).view(_trajectory_dtype())
    out = np.concatenate([everything[r * per_rank:r * per_rank + (shard_range(int(num_games), r, world)[1]
                                                                  - shard_range(int(num_games), r, world)[0])]
                          for r in range(world)])
    return out, stats


def _trajectory_dtype():
    from .engine import TRAJECTORY_DTYPE
    return TRAJECTORY_DTYPE


def _planes(b1: int, b2: int, me: int) -> np.ndarray:
    board = bits_to_board(b1, b2)
    rival = 2 if me == 1 else 1
    return np.stack([board == me, board == rival]).astype(np.float32)


def self_play(agent, num_games) -> List[tuple]:
    """train_alpha_zero.py:33-55 on the GPU: one sample per game - the final position seen from
    the side to move after the last stone (agent.current_player there), a zero policy vector,
    and get_result() in colour 1's frame."""
    if agent.load_model_flag and not agent.model_loaded:
        agent.load_model()
    records, stats = play_games(agent, int(num_games))
    memory = []
    for r in records:
        n = int(r["length"])
        mover_after = 1 if n == 0 else 3 - int(r["plies"]["player"][n - 1])
        agent.current_player = mover_after
        planes = torch.from_numpy(_planes(int(r["final_boards"][0]), int(r["final_boards"][1]), mover_after))
        result = None if r["result"] == 2 else int(r["result"])
        memory.append((planes.to(agent.device), torch.zeros(agent.action_dim, dtype=torch.float32), result))
    logger.info("self-play: %d games, %d plies, %d simulations", stats.get("games", 0), stats.get("plies", 0),
                stats.get("sims", 0))
    agent.memory.extend(memory)
    return memory


def self_play_samples(agent, num_games) -> List[tuple]:
    """Per-ply training samples (planes[2,6,7], visit distribution[7], z) as
    agent_code.py:254-267 intends: z is +1 when the side to move at that ply went on to win,
    -1 when it lost, 0 for a draw."""
    records, _ = play_games(agent, int(num_games))
    out = []
    for r in records:
        n = int(r["length"])
        res = int(r["result"])
        for i in range(n):
            ply = r["plies"][i]
            me = int(ply["player"])
            vis = ply["visits"].astype(np.float64)
            probs = vis / vis.sum() if vis.sum() > 0 else vis
            z = 0 if res in (0, 2) else (1 if (res == 1) == (me == 1) else -1)
            out.append((torch.from_numpy(_planes(int(ply["colour_boards"][0]), int(ply["colour_boards"][1]), me)),
                        torch.from_numpy(probs.astype(np.float32)), z))
    return out


def self_play_sample_tensors(agent, num_games, first_game_id: int = 0):
    """self_play_samples without the host loop: the games are played and their per-ply samples built on the
    device (nnb_trajectory_samples); returns (planes [P,2,6,7], visit distributions [P,7], z [P]) as device
    tensors in the order self_play_samples lists them - what ConnectFourDataset (data_module.py:13-27) stacks."""
    num_games = int(num_games)
    eng = agent.engine(min(max(num_games, 1), MAX_CONCURRENT_GAMES))
    params = eng.params(agent.num_simulations, agent.c_puct, evaluator=agent.evaluator, mode="compat",
                        eval_cache=getattr(agent, "eval_cache", False))
    traj, _ = eng.selfplay_round(num_games, params, first_game_id=first_game_id)
    return eng.trajectory_samples(traj)


def train_alphazero(time_limit, num_self_play_games=100, use_gpu=True, load_model=True):
    """Entry point kept for import compatibility (train_alpha_zero.py:57-119).  The self-play half
    runs on the engine; the Lightning training half is not part of this package."""
    from .agent import initialize_agent
    agent = initialize_agent(action_dim=7, state_dim=2, use_gpu=use_gpu, num_simulations=800, c_puct=1.4,
                             load_model=load_model)
    self_play(agent, num_self_play_games)
    raise NotImplementedError("self-play finished and agent.memory is filled; the pytorch_lightning trainer of "
                              "the reference (train_alpha_zero.py:79-115) is outside this package")

"""Multi-rank path on CPU: two gloo ranks shard a round of games by index, each fabricates the
trajectory records of its shard, and the all-gather puts every record on every rank in game
order - the same code path bench.py and selfplay.gather_trajectories use over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alphazero_amd.engine import TRAJECTORY_DTYPE
from alphazero_amd.selfplay import gather_trajectories, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_records(first, last):
    rec = np.zeros(last - first, dtype=TRAJECTORY_DTYPE)
    for i, gid in enumerate(range(first, last)):
        rec[i]["game_id"] = gid
        rec[i]["length"] = 7 + gid % 5
        rec[i]["result"] = (gid % 3) - 1
        rec[i]["final_boards"] = (gid * 3 + 1, gid * 5 + 2)
        rec[i]["plies"]["action"][:] = gid % 7
        rec[i]["plies"]["visits"][:, 0] = gid % 200
    return rec


def _worker(rank, world, port, n_games, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, last = shard_range(n_games, rank, world)
        local = torch.from_numpy(_fake_records(first, last).view(np.uint8).reshape(last - first, -1).copy())
        everything = gather_trajectories(local)
        rec = everything.numpy().reshape(-1).view(TRAJECTORY_DTYPE)
        want = _fake_records(0, n_games)
        assert rec.shape == want.shape
        assert (rec.view(np.uint8) == want.view(np.uint8)).all()
        # barrier + max-over-ranks timing reduction used by the benchmark
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_trajectory_allgather(tmp_path):
    world, n_games = 2, 64
    mp.spawn(_worker, args=(world, _free_port(), n_games, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _worker_uneven(rank, world, port, n_games, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alphazero_amd import selfplay
        selfplay.play_games = lambda agent, n, first_game_id=0, **kw: (_fake_records(first_game_id, first_game_id + n),
                                                                       {"games": n})
        rec, stats = selfplay.distributed_play_games(object(), n_games, first_game_id=100)
        want = _fake_records(100, 100 + n_games)
        assert rec.shape == want.shape and rec.tobytes() == want.tobytes()
        first, last = shard_range(n_games, rank, world)
        assert stats == {"games": last - first}
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_distributed_play_games_with_uneven_shares(tmp_path):
    world, n_games = 2, 7                      # shares of 4 and 3 games
    mp.spawn(_worker_uneven, args=(world, _free_port(), n_games, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_gather_is_identity_without_a_process_group():
    local = torch.zeros((3, TRAJECTORY_DTYPE.itemsize), dtype=torch.uint8)
    assert gather_trajectories(local) is local


def test_selfplay_memory_from_records(monkeypatch):
    """self_play() turns trajectory records into the reference's per-game samples."""
    from alphazero_amd import selfplay
    from alphazero_amd.agent import initialize_agent
    from oracle import c_oracle as co
    from oracle import ref_port as rp

    recs = np.zeros(3, dtype=TRAJECTORY_DTYPE)
    want = []
    for i, ev in enumerate(("const", "hash_q8", "hash_f32")):
        moves, visits, result, final, _ = co.selfplay_game(60, 1.4, ev)
        recs[i]["length"] = len(moves)
        recs[i]["result"] = result
        recs[i]["final_boards"] = final
        player = 1
        for k, m in enumerate(moves):
            recs[i]["plies"][k]["action"] = m
            recs[i]["plies"][k]["player"] = player
            recs[i]["plies"][k]["visits"] = visits[k]
            player = 3 - player
        want.append((final, player, result))
    monkeypatch.setattr(selfplay, "play_games", lambda agent, n, **kw: (recs[:n], {"games": n}))
    agent = initialize_agent(load_model=False, num_simulations=60)
    mem = selfplay.self_play(agent, 3)
    assert len(mem) == 3 and agent.memory == mem
    for (planes, probs, result), (final, mover, res) in zip(mem, want):
        assert tuple(planes.shape) == (2, 6, 7) and planes.dtype == torch.float32
        assert tuple(probs.shape) == (7,) and float(probs.abs().sum()) == 0.0
        assert result == res
        board = np.zeros((6, 7), dtype=np.int8)
        for c in range(7):
            for r in range(6):
                if final[0] >> (c * 7 + r) & 1:
                    board[5 - r][c] = 1
                if final[1] >> (c * 7 + r) & 1:
                    board[5 - r][c] = 2
        assert (planes.numpy() == rp.planes(board, mover)).all()
    samples = selfplay.self_play_samples(agent, 3)
    assert len(samples) == int(recs["length"].sum())
    assert all(abs(float(p.sum()) - 1.0) < 1e-6 for _, p, _ in samples)
    assert {z for _, _, z in samples} <= {-1, 0, 1}

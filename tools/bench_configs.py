"""Throughput of the workloads other than the headline one (bench.py measures that):

  tree     pure-tree search, PUCT with uniform priors + bitboard random rollouts, 800 sims/move
           (whole move in one launch, node pool in LDS); also the reference rule with the same evaluator
  az       PUCT + Dirichlet root noise with the network, 4096 games x 200 sims/move
  cache    the headline workload and PUCT self-play with the per-game evaluation cache on and off
  diverse  the reference rule with the network, 4 random opening plies, 4 x as many games as slots
  arena    two networks against each other, 400 sims/move, starter alternates, both nets batched

  python tools/bench_configs.py [tree az diverse arena] [--games N]

Prints one JSON line per workload.  Inputs are resident on the device before the clock starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alphazero_amd.engine import Engine, trajectories_to_numpy  # noqa: E402
from alphazero_amd.network import seeded_state_dict  # noqa: E402


def timed_round(eng, n, params, starters=None, reps=1, total=None):
    """`total` games (default n) on an engine of n slots."""
    zeros = np.zeros((n, 2), dtype=np.uint64) if starters is not None else None
    eng.selfplay_round(min(n, 256), params)                     # warm-up (code objects, clocks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = {}
    for r in range(reps):
        traj, st = eng.selfplay_round(total or n, params, first_game_id=r * n, start_boards=zeros, start_players=starters)
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    torch.cuda.synchronize()
    return time.perf_counter() - t0, tot, traj


def report(name, dt, st, extra=None):
    out = {"workload": name, "games_per_s": st["games"] / dt, "sims_per_s": st["sims"] / dt, "seconds": dt,
           "games": st["games"], "plies_per_game": st["plies"] / max(st["games"], 1),
           "mean_selection_depth": st["depth_sum"] / max(st["sims"], 1)}
    out.update(extra or {})
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["tree", "az", "diverse", "arena"])
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--arena-games", type=int, default=10000)
    a = ap.parse_args()
    G = a.games
    if "tree" in a.what:
        eng = Engine(max_games=G, max_sims=800)
        for mode in ("az", "compat"):
            p = eng.params(800, 1.4, evaluator="rollout", mode=mode, seed=7)
            dt, st, _ = timed_round(eng, G, p)
            report(f"tree-only {mode}: {G} games x 800 sims/move, uniform priors + random rollouts", dt, st)
        eng.close()
    if "az" in a.what:
        eng = Engine(max_games=G, max_sims=200)
        eng.load_state_dict(seeded_state_dict(0))
        p = eng.params(200, 1.4, evaluator="net", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=11)
        for total in (G, 4 * G):
            eng.profile(True)
            dt, st, _ = timed_round(eng, G, p, total=total)
            prof = {k: round(v["avg_ms"], 4) for k, v in eng.profile_report().items()}
            eng.profile(False)
            print(json.dumps({"kernel_ms": prof, "steps": st["steps"], "moves_rounds": st["moves_rounds"],
                              "net_leaves": st["net_leaves"], "terminal_hits": st["terminal_hits"]}))
            report(f"az: {total} games on {G} slots x 200 sims/move, PUCT + Dirichlet(0.3, eps 0.25), network", dt, st,
                   {"leaf_batch_fill": st["net_leaves"] / max(st["steps"] * G, 1)})
        eng.close()
    if "cache" in a.what:
        eng = Engine(max_games=G, max_sims=200, eval_cache_slots=int(os.environ.get("NNB_CACHE_SLOTS", "2048")))
        eng.load_state_dict(seeded_state_dict(0))
        for mode, extra, total in (("compat", {}, G), ("az", dict(dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=11), 4 * G)):
            for cache in (False, True):
                p = eng.params(200, 1.4, evaluator="net", mode=mode, eval_cache=cache, **extra)
                eng.profile(True)
                dt, st, _ = timed_round(eng, G, p, total=total)
                print(json.dumps({"kernel_ms": {k: round(v["avg_ms"], 4) for k, v in eng.profile_report().items()}}))
                eng.profile(False)
                report(f"{mode}, network, {total} games on {G} slots x 200 sims/move, evaluation cache {'on' if cache else 'off'}",
                       dt, st, {"net_leaves": st["net_leaves"], "cache_hits": st["cache_hits"], "steps": st["steps"]})
        eng.close()
    if "diverse" in a.what:
        from alphazero_amd.selfplay import random_openings
        eng = Engine(max_games=G, max_sims=200)
        eng.load_state_dict(seeded_state_dict(0))
        p = eng.params(200, 1.4, evaluator="net", mode="compat")
        total = 4 * G
        sb, sp = random_openings(eng, total, 4, seed=1)
        eng.selfplay_round(256, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        traj, st = eng.selfplay_round(total, p, start_boards=sb, start_players=sp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        report(f"diverse: {total} games on {G} slots x 200 sims/move, reference rule, 4 random opening plies, network", dt, st,
               {"leaf_batch_fill": st["net_leaves"] / max(st["steps"] * G, 1)})
        eng.close()
    if "arena" in a.what:
        eng = Engine(max_games=G, max_sims=400, n_weight_sets=2)
        eng.load_state_dict(seeded_state_dict(0), 0)
        eng.load_state_dict(seeded_state_dict(1), 1)
        p = eng.params(400, 1.4, evaluator="net", mode="compat", weight_set=0, weight_set_colour2=1)
        eng.selfplay_round(256, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = a.arena_games
        starters = (1 + (np.arange(n) % 2)).astype(np.int8)
        traj, tot = eng.selfplay_round(n, p, start_boards=np.zeros((n, 2), dtype=np.uint64), start_players=starters)
        res = trajectories_to_numpy(traj)["result"].copy()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        report(f"arena: {a.arena_games} games, two networks (seeds 0 and 1), 400 sims/move, {G} slots refilled", dt, tot,
               {"wins_colour1": int((res == 1).sum()), "wins_colour2": int((res == -1).sum()),
                "draws": int(((res != 1) & (res != -1)).sum())})
        eng.close()


if __name__ == "__main__":
    main()

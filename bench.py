#!/usr/bin/env python3
"""Self-play throughput benchmark: 4096 concurrent games x 200 sims/move per GPU (one GPU at N=1,
the same per-GPU share on each of N GPUs at N>1).

One "step" = one self-play round: every rank plays GAMES_PER_GPU concurrent Connect Four
games to the end with the reference's search (200 simulations per move, same searcher
on both sides, random-init seed-0 Connect4Net as the leaf evaluator, all leaves of a
simulation round batched into one forward pass), then the finished trajectories are
all-gathered over RCCL.  Weak scaling: per-GPU work is fixed.

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel's algorithmic FLOP rate against the MI355X dense matrix peak of
                 the arithmetic in use (f16 inputs: 2.5 PFLOP/s; f32: 157.3 TFLOP/s),
                 its duration measured with HIP events on the launch stream inside this run;
  cpu_baseline : the CPU restatement of the reference path (oracle/ref_port.py: numpy board,
                 one game at a time, batch-1 torch-CPU forward) timed on this box's host cores
                 on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GAMES_PER_GPU = 4096
SIMS = 200
C_PUCT = 1.4

# algorithmic FLOP per evaluated leaf (2 x MAC), network.py:28-41
FLOP_PER_LEAF = {
    "conv0": 2 * 128 * 7 * 8 * (2 * 16),
    "conv2": 2 * 128 * 8 * 9 * (128 * 16),
    "conv4": 2 * 128 * 9 * 10 * (128 * 16),
    "fc": 2 * 11520 * 1024,
    "heads": 2 * 1024 * 8,
}
# MI355X dense matrix peaks of the MFMA instruction each path issues.  "f32s" issues half-input MFMAs, three
# per float32 product; its FLOPs are counted once per product, so at best a third of that peak is reachable.
PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f32s": 2500.0}
DTYPE_NAME = {"f16": "f16 (f32 accumulate)", "f32": "f32",
              "f32s": "f32 (each operand a hi+lo pair of halves, 3 half-input MFMAs per product, f32 accumulate)"}
PEAK_HBM_GBS = 8000.0


def cpu_baseline(sims: int, plies_per_game: float, budget_s: float = 25.0) -> dict:
    """Times the CPU restatement of the reference self-play path on a bounded sample."""
    import torch
    from oracle import ref_port as rp
    sd = rp.seeded_state_dict(0)
    ev = rp.make_net_evaluator(sd)
    board, player = rp.new_board(), 1
    t0 = time.perf_counter()
    plies = 0
    done = False
    while not rp.is_terminal(board):
        a, _ = rp.compat_search(board, player, sims, C_PUCT, ev)
        _, player = rp.make_move(board, player, a)
        plies += 1
        if time.perf_counter() - t0 > budget_s:
            break
    else:
        done = True
    dt = time.perf_counter() - t0
    sims_per_s = plies * sims / dt
    if done:
        games_per_s = 1.0 / dt
        sample = f"1 complete game, {plies} plies x {sims} sims, {dt:.1f} s"
    else:
        games_per_s = sims_per_s / (sims * max(plies_per_game, 1.0))
        sample = (f"first {plies} plies x {sims} sims of one game in {dt:.1f} s; games/s extrapolated to "
                  f"{plies_per_game:.1f} plies per game")
    return {"value": games_per_s, "unit": "games/s", "sims_per_s": sims_per_s,
            "cores": torch.get_num_threads(), "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=SIMS)
    ap.add_argument("--precision", choices=["f16", "f32", "f32s"], default="f32s",
                    help="arithmetic of the network's conv / linear layers (both accumulate in f32)")
    ap.add_argument("--opening-plies", type=int, default=0,
                    help="random legal moves played before self-play starts (0 = every game from the empty board, as the\n"
                         "reference plays; its search is deterministic, so those games are all the same game)")
    ap.add_argument("--no-eval-cache-line", action="store_true",
                    help="skip the extra, untimed round with the per-game evaluation cache (reported apart from `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-stream", action="store_true",
                    help="keep all games in one tree step -> forward pass chain instead of two halves on two streams")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from alphazero_amd.engine import TRAJECTORY_BYTES, Engine
    from alphazero_amd.network import seeded_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # Under a launcher (RANK set) the collective path runs even with one rank, so the single-GPU
    # launch of `python -m torch.distributed.run --nproc-per-node 1 bench.py` exercises RCCL too.
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout is
        # reserved for the one JSON line, so the banner goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=device)
            dist.barrier()
            torch.cuda.synchronize(device)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    G = args.games
    cache_line = not args.no_eval_cache_line
    engine = Engine(max_games=G, max_sims=args.sims, device=device, precision=args.precision,
                    eval_cache_slots=2048 if cache_line else 0)
    engine.load_state_dict(seeded_state_dict(0))
    params = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", one_stream=args.one_stream)
    traj = torch.empty((G, TRAJECTORY_BYTES), dtype=torch.uint8, device=device)
    replay = torch.empty((world * G, TRAJECTORY_BYTES), dtype=torch.uint8, device=device) if use_dist else traj

    start_boards = start_players = None
    if args.opening_plies > 0:
        from alphazero_amd.selfplay import random_openings
        start_boards, start_players = random_openings(engine, G, args.opening_plies, seed=1234 + rank)

    def one_round(step_idx: int):
        _, stats = engine.selfplay_round(G, params, first_game_id=(step_idx * world + rank) * G, out=traj,
                                         start_boards=start_boards, start_players=start_players)
        if use_dist:
            dist.all_gather_into_tensor(replay, traj)      # RCCL: trajectories into every rank's buffer
        return stats

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    for w in range(args.warmup):
        one_round(w)
    engine.profile(True)
    fence()
    t0 = time.perf_counter()
    totals = {}
    for k in range(args.steps):
        st = one_round(args.warmup + k)
        for key, v in st.items():
            totals[key] = totals.get(key, 0) + v
    fence()
    elapsed = time.perf_counter() - t0
    prof = engine.profile_report()
    engine.profile(False)

    # One more round, outside the timed region and apart from `value`: the same games with the per-game
    # evaluation cache (a leaf whose position the game has already evaluated is not sent to the
    # network again).  The reference evaluates every leaf, so the headline number does too.
    cached = None
    if cache_line:
        cparams = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", eval_cache=True)
        fence()
        c0 = time.perf_counter()
        _, cst = engine.selfplay_round(G, cparams, first_game_id=0, out=traj, start_boards=start_boards,
                                       start_players=start_players)
        torch.cuda.synchronize(device)
        cdt = time.perf_counter() - c0
        cached = {"games_per_s_this_rank": cst["games"] / cdt, "sims_per_s_this_rank": cst["sims"] / cdt,
                  "net_leaves": cst["net_leaves"], "cache_hits": cst["cache_hits"],
                  "note": "same trajectories; not part of `value`"}

    # all-gather cost alone (reported, also part of the timed region above)
    gather_ms = None
    if use_dist:
        fence()
        g0 = time.perf_counter()
        dist.all_gather_into_tensor(replay, traj)
        torch.cuda.synchronize(device)
        gather_ms = (time.perf_counter() - g0) * 1e3

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        keys = sorted(totals)
        s = torch.tensor([totals[k] for k in keys], dtype=torch.int64, device=device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        job = {k: int(v) for k, v in zip(keys, s.tolist())}
    else:
        job = dict(totals)

    if rank == 0:
        games = job["games"]
        steps_launched = max(totals["steps"], 1)
        leaves_per_launch = totals["net_leaves"] / steps_launched          # this rank
        # dominant kernel = largest share of sampled device time
        shares = {k: v["avg_ms"] for k, v in prof.items()}
        dom = max(FLOP_PER_LEAF, key=lambda k: shares.get(k, 0.0))
        avg_ms = prof[dom]["avg_ms"]
        achieved = (leaves_per_launch * FLOP_PER_LEAF[dom]) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        net_ms = sum(prof[k]["avg_ms"] for k in FLOP_PER_LEAF)
        net_tflops = (leaves_per_launch * sum(FLOP_PER_LEAF.values())) / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
        traffic, traffic_src = None, None
        try:      # HBM bytes per launch from the committed rocprofv3 --pmc passes (collected offline)
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
                tj = json.load(open(path))
                if tj.get("dtype", "f32") == args.precision and G == GAMES_PER_GPU and args.sims == SIMS:
                    traffic = tj["kernels"][dom]["hbm_bytes"]
                    traffic_src = os.path.relpath(path, ROOT)
                    break
        except Exception:
            traffic = None
        plies_per_game = job["plies"] / max(games, 1)
        depth_mean = job["depth_sum"] / max(job["sims"], 1)
        tree_bytes_per_sim = 112.0 * depth_mean + 184.0
        tree_ms = prof["tree_step"]["avg_ms"]
        sims_per_launch = totals["sims"] / max(totals["steps"] + totals["moves_rounds"], 1)
        tree_gbs = (sims_per_launch * tree_bytes_per_sim) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
        out = {
            "metric": "selfplay_games_per_s",
            "value": games / elapsed,
            "unit": "games/s",
            "sims_per_s": job["sims"] / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision],
            "data": "synthetic: games from the empty board, random-init Connect4Net (torch.manual_seed(0))",
            "config": {"workload": f"{G} concurrent games per GPU x {args.sims} sims/move, batched leaf inference, "
                                   "reference search rule (compat), every game played to the end"
                                   + (f", {args.opening_plies} random opening plies" if args.opening_plies else ""),
                       "games_per_gpu": G, "sims_per_move": args.sims, "c_puct": C_PUCT,
                       "parallelism": f"games sharded over {world} GPU(s), no collective inside the search; "
                                      "one all-gather of trajectories per round"},
            "stats": {"games": games, "plies": job["plies"], "sims": job["sims"], "net_leaves": job["net_leaves"],
                      "terminal_hits": job["terminal_hits"], "plies_per_game": plies_per_game,
                      "mean_selection_depth": depth_mean,
                      "leaf_batch_fill": job["net_leaves"] / max(job["steps"] * G, 1),
                      "allgather_ms": gather_ms,
                      "allgather_bytes": world * G * TRAJECTORY_BYTES if use_dist else 0},
            "kernel_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()},
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.precision],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic,
                         "traffic_source": traffic_src,
                         "flop_per_leaf": FLOP_PER_LEAF[dom], "leaves_per_launch": leaves_per_launch,
                         "avg_launch_ms": avg_ms, "whole_net_tflops": net_tflops},
            "roofline_tree": {"kernel": "tree_step", "bound": "hbm", "achieved": tree_gbs, "peak": PEAK_HBM_GBS,
                              "unit": "GB/s", "frac": tree_gbs / PEAK_HBM_GBS, "traffic": None,
                              "bytes_per_sim": tree_bytes_per_sim, "avg_launch_ms": tree_ms},
        }
        if cached is not None:
            out["with_eval_cache"] = cached
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims, plies_per_game)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Self-play throughput benchmark: 4096 concurrent games x 200 sims/move per GPU (one GPU at N=1,
the same per-GPU share on each of N GPUs at N>1).

One "step" = one self-play round: every rank plays GAMES_PER_STEP Connect Four games to the end,
GAMES_PER_GPU of them at a time (a finished game's slot takes the round's next game), with the
reference's search (200 simulations per move, same searcher on both sides, random-init seed-0
Connect4Net in float32-faithful arithmetic as the leaf evaluator, all leaves of a simulation round
batched into one forward pass), then the finished trajectories are all-gathered over RCCL.
The reference's search is deterministic, so its games from the empty board are all the same game;
the timed games therefore start from distinct positions (OPENING_PLIES uniformly random legal moves).
Weak scaling: per-GPU work is fixed.

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel's algorithmic FLOP rate (each float32 product counted once) against
                 the MI355X dense peak of the matrix instruction it issues (half inputs: 2.5 PFLOP/s;
                 exact f32: 157.3 TFLOP/s); its duration is sampled with HIP events on the stream the
                 kernel is launched on, inside the timed region (16 launches of every 64 lock-step rounds);
  cpu_baseline : the CPU restatement of the reference path (oracle/ref_port.py: numpy board,
                 one game at a time, batch-1 torch-CPU forward) timed on this box's host cores
                 on a bounded sample, at 200 and at 100 simulations per move (rank 0, N=1 only);
  single_round_identical_games, f16_inputs, with_eval_cache (N=1 only, each timed on its own, none part
                 of `value`): the 4096 games from the empty board the reference itself would play (one game,
                 4096 times); the headline workload with half-precision network inputs; and with the
                 per-game evaluation cache.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GAMES_PER_GPU = 4096          # concurrent games (slots)
GAMES_PER_STEP = 16384        # games per self-play round and GPU
OPENING_PLIES = 4
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


def cpu_baseline(sims: int, budget_s: float = 20.0) -> dict:
    """Times the CPU restatement of the reference self-play path (one game from the empty board, as
    train_alpha_zero.py:33-55 plays it) on a bounded sample."""
    import torch
    from oracle import ref_port as rp
    sd = rp.seeded_state_dict(0)
    ev = rp.make_net_evaluator(sd)
    board, player = rp.new_board(), 1
    plies_per_game = 14.0 if sims == 200 else 24.0      # length of that game when the budget cuts it short
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


def committed_traffic(precision: str, kernel: str, leaves_per_launch: float):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary of this command
    (collected offline, separate FETCH_SIZE / WRITE_SIZE passes), or (None, None) when the summary on
    file was taken at another precision or batch size per launch."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(path))
            if tj.get("dtype", "f32") != precision or kernel not in tj["kernels"]:
                continue
            lpl = tj.get("leaves_per_launch")
            if lpl is not None and abs(lpl - leaves_per_launch) > 0.1 * max(leaves_per_launch, 1.0):
                continue
            return tj["kernels"][kernel]["hbm_bytes"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games (slots) per GPU")
    ap.add_argument("--games-per-step", type=int, default=0,
                    help=f"games per round and GPU (default {GAMES_PER_STEP // GAMES_PER_GPU} x --games)")
    ap.add_argument("--sims", type=int, default=SIMS)
    ap.add_argument("--precision", choices=["f16", "f32", "f32s"], default="f32s",
                    help="arithmetic of the network's conv / linear layers (all accumulate in f32); f32s and f32 are\n"
                         "float32-faithful, f16 rounds the inputs of every product to half precision")
    ap.add_argument("--opening-plies", type=int, default=OPENING_PLIES,
                    help="random legal moves played before self-play starts (0 = every game from the empty board, as the\n"
                         "reference plays; its search is deterministic, so those games are all the same game)")
    ap.add_argument("--no-extra-lines", action="store_true",
                    help="skip the extra rounds reported apart from `value` (identical games, half-precision inputs, evaluation cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-stream", action="store_true",
                    help="keep all games in one tree step -> forward pass chain instead of two halves on two streams")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from alphazero_amd.engine import TRAJECTORY_BYTES, Engine
    from alphazero_amd.network import seeded_state_dict
    from alphazero_amd.selfplay import random_openings

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

    G = args.games                                             # slots
    NG = args.games_per_step if args.games_per_step > 0 else G * (GAMES_PER_STEP // GAMES_PER_GPU)
    extras = world == 1 and not args.no_extra_lines
    engine = Engine(max_games=G, max_sims=args.sims, device=device, precision=args.precision,
                    eval_cache_slots=2048 if extras else 0)
    engine.load_state_dict(seeded_state_dict(0))
    params = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", one_stream=args.one_stream)
    traj = torch.empty((NG, TRAJECTORY_BYTES), dtype=torch.uint8, device=device)
    replay = torch.empty((world * NG, TRAJECTORY_BYTES), dtype=torch.uint8, device=device) if use_dist else traj

    def openings(step_idx: int):
        if args.opening_plies <= 0:
            return None, None
        b, p = random_openings(engine, NG, args.opening_plies, seed=1234 + step_idx * world + rank)
        return (torch.from_numpy(b.view("int64")).to(device), torch.from_numpy(p).to(device))

    def one_round(eng, prm, step_idx: int, start, n_games=NG, out=traj):
        _, stats = eng.selfplay_round(n_games, prm, first_game_id=(step_idx * world + rank) * NG, out=out[:n_games],
                                      start_boards=start[0], start_players=start[1])
        if use_dist and eng is engine and n_games == NG:
            dist.all_gather_into_tensor(replay, traj)      # RCCL: trajectories into every rank's buffer
        return stats

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    # start positions are inputs: resident in HBM before the timed region starts
    starts = [openings(i) for i in range(args.warmup + args.steps)]
    for w in range(args.warmup):
        one_round(engine, params, w, starts[w])
    engine.profile(True)
    fence()
    t0 = time.perf_counter()
    totals = {}
    for k in range(args.steps):
        st = one_round(engine, params, args.warmup + k, starts[args.warmup + k])
        for key, v in st.items():
            totals[key] = v if key == "chains" else totals.get(key, 0) + v
    fence()
    elapsed = time.perf_counter() - t0
    prof = engine.profile_report()
    engine.profile(False)

    def timed_extra(eng, prm, start, n_games, note):
        fence()
        c0 = time.perf_counter()
        st = one_round(eng, prm, 0, start, n_games=n_games)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - c0
        return {"games_per_s": st["games"] / dt, "sims_per_s": st["sims"] / dt, "seconds": dt, "games": st["games"],
                "plies_per_game": st["plies"] / max(st["games"], 1), "net_leaves": st["net_leaves"],
                "cache_hits": st["cache_hits"], "note": note}

    # Extra rounds on one GPU, each timed on its own and none part of `value`.
    extra = {}
    if extras:
        last = starts[-1]
        extra["single_round_identical_games"] = timed_extra(
            engine, params, (None, None), G,
            f"{G} games from the empty board in one round with no refill: the reference's search is deterministic, "
            "so this is one game played on every slot (best-case batch fill)")
        cparams = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", eval_cache=True)
        extra["with_eval_cache"] = timed_extra(
            engine, cparams, last, NG,
            "the headline games with the per-game evaluation cache: a leaf whose position its game has already "
            "evaluated is not sent to the network again; same trajectories; the reference evaluates every leaf")
        if args.precision != "f16":
            e16 = Engine(max_games=G, max_sims=args.sims, device=device, precision="f16")
            e16.load_state_dict(seeded_state_dict(0))
            p16 = e16.params(args.sims, C_PUCT, evaluator="net", mode="compat")
            one_round(e16, p16, 0, last)
            extra["f16_inputs"] = timed_extra(
                e16, p16, last, NG,
                "the headline workload with the inputs of every product rounded to half precision (f32 accumulate): "
                "narrower than the reference's float32, so not comparable with `value`")
            e16.close()

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
        chains = max(totals.get("chains", 1), 1)       # 2: each lock-step round is two launches per kernel, half the slots each
        steps_launched = max(totals["steps"], 1) * chains
        leaves_per_launch = totals["net_leaves"] / steps_launched          # this rank
        # dominant kernel = largest share of sampled device time
        shares = {k: v["avg_ms"] for k, v in prof.items()}
        dom = max(FLOP_PER_LEAF, key=lambda k: shares.get(k, 0.0))
        avg_ms = prof[dom]["avg_ms"]
        achieved = (leaves_per_launch * FLOP_PER_LEAF[dom]) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        net_ms = sum(prof[k]["avg_ms"] for k in FLOP_PER_LEAF)
        net_tflops = (leaves_per_launch * sum(FLOP_PER_LEAF.values())) / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
        standard = G == GAMES_PER_GPU and args.sims == SIMS and NG == GAMES_PER_STEP and args.opening_plies == OPENING_PLIES
        traffic, traffic_src = committed_traffic(args.precision, dom, leaves_per_launch) if standard else (None, None)
        tree_traffic, _ = committed_traffic(args.precision, "tree_step", leaves_per_launch) if standard else (None, None)
        plies_per_game = job["plies"] / max(games, 1)
        depth_mean = job["depth_sum"] / max(job["sims"], 1)
        tree_bytes_per_sim = 112.0 * depth_mean + 184.0
        tree_ms = prof["tree_step"]["avg_ms"]
        sims_per_launch = totals["sims"] / steps_launched
        tree_gbs = (sims_per_launch * tree_bytes_per_sim) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
        mfma_per_product = 3 if args.precision == "f32s" else 1
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
            "data": "synthetic: " + (f"start positions after {args.opening_plies} uniformly random legal plies"
                                     if args.opening_plies else "games from the empty board")
                    + ", random-init Connect4Net (torch.manual_seed(0))",
            "config": {"workload": f"{G} concurrent games per GPU x {args.sims} sims/move, batched leaf inference, "
                                   f"reference search rule (compat), {NG} games per round and GPU played to the end "
                                   "(a finished game's slot takes the next game)",
                       "games_per_gpu": G, "games_per_step_per_gpu": NG, "sims_per_move": args.sims, "c_puct": C_PUCT,
                       "opening_plies": args.opening_plies,
                       "parallelism": f"games sharded over {world} GPU(s), no collective inside the search; "
                                      "one all-gather of trajectories per round"},
            "stats": {"games": games, "plies": job["plies"], "sims": job["sims"], "net_leaves": job["net_leaves"],
                      "terminal_hits": job["terminal_hits"], "plies_per_game": plies_per_game,
                      "mean_selection_depth": depth_mean,
                      "leaf_batch_fill": job["net_leaves"] / max(job["steps"] * G, 1),
                      "lockstep_rounds": job["steps"], "launch_chains_per_round": chains,
                      "allgather_ms": gather_ms,
                      "allgather_bytes": world * NG * TRAJECTORY_BYTES if use_dist else 0},
            "kernel_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()},
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.precision],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic,
                         "traffic_source": traffic_src,
                         "flop_per_leaf": FLOP_PER_LEAF[dom], "leaves_per_launch": leaves_per_launch,
                         "avg_launch_ms": avg_ms, "whole_net_tflops": net_tflops,
                         "mfma_per_product": mfma_per_product,
                         "issued_frac": achieved * mfma_per_product / PEAK_TFLOPS[args.precision]},
            "roofline_tree": {"kernel": "tree_step", "bound": "hbm", "achieved": tree_gbs, "peak": PEAK_HBM_GBS,
                              "unit": "GB/s", "frac": tree_gbs / PEAK_HBM_GBS, "traffic": tree_traffic,
                              "bytes_per_sim": tree_bytes_per_sim, "sims_per_launch": sims_per_launch,
                              "avg_launch_ms": tree_ms},
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims)
            out["cpu_baseline"]["config1_100_sims"] = cpu_baseline(100, budget_s=10.0)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()

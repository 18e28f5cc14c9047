#!/usr/bin/env python3
"""Self-play throughput benchmark: 4096 concurrent games x 200 sims/move per GPU (one GPU at N=1,
the same per-GPU share on each of N GPUs at N>1).

Every rank plays an endless stream of Connect Four self-play games on GAMES_PER_GPU slots (a finished
game's slot takes the next game, so the leaf batch stays full) with the reference's search (200
simulations per move, same searcher on both sides, random-init seed-0 Connect4Net in float32-faithful
arithmetic as the leaf evaluator, all leaves of a simulation round batched into one forward pass).
One "step" = wait until GAMES_PER_GPU more games have finished on this rank, then all-gather the step's
finished trajectories over RCCL.  `value` = games finished inside the timed region / its duration: games in
flight when the region starts were begun by the warm-up steps, as those in flight at its end are finished
by whatever follows - the steady state of a self-play service.  `sims_per_s` counts the searches completed
inside the region, whichever game they belong to.  (A plain round of games played until the last one has
ended is reported as single_round_*.)
The reference's search is deterministic, so its games from the empty board are all the same game;
the timed games therefore start from OPENING_PLIES uniformly random legal moves (16384 drawn start
positions per rank; 1119 of them are distinct).  Weak scaling: per-GPU work is fixed.

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel's algorithmic FLOP rate (each float32 product counted once) against
                 the MI355X dense peak of the matrix instruction it issues (half inputs: 2.5 PFLOP/s;
                 exact f32: 157.3 TFLOP/s); its duration is sampled with HIP events on the stream the
                 kernel is launched on, inside the timed region (16 launches of every 64 lock-step rounds).
                 The timed region runs two chains on two streams, where a launch shares the chip with the other
                 chain's kernel: those durations are reported as roofline_timed_region (with region_frac, the
                 network's algorithmic FLOPs of the whole timed region over its wall time), and `roofline` itself
                 comes from one more step, right after the timed region, with all games in ONE chain (N=1 only;
                 with --one-stream or at N>1 `roofline` is measured inside the timed region);
  cpu_baseline : the CPU restatement of the reference path (oracle/ref_port.py: numpy board,
                 one game at a time, batch-1 torch-CPU forward) timed on this box's host cores
                 on a bounded sample, at 200 and at 100 simulations per move (rank 0, N=1 only);
  single_round_identical_games, single_round_distinct_games, with_eval_cache, az_mode, f16_inputs, tree_only_800_sims (N=1 only,
                 each timed on its own, none part of `value`): the 4096 games from the empty board the reference
                 itself would play (one game, 4096 times) as one plain round; 16384 of the timed region's games as
                 one plain round; the stream with the per-game evaluation cache; a stream under the PUCT rule with
                 Dirichlet root noise; the headline stream with half-precision network inputs; and the tree kernels
                 without the network (800 simulations per move, random play-outs: BASELINE.json's pure-tree configuration).
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
# What the matrix pipes of one box deliver from registers with changing operands and nothing else running: the socket's
# power cap holds the clock at 1.62 GHz (tools/mfma_peak, profiles/r03_mfma_power_ceiling.txt).  Reported beside `peak`.
POWER_CAPPED_MFMA_TFLOPS = 1630.0
KERNEL_NAME = {}


def cpu_baseline(sims: int, budget_s: float = 30.0, games: int = 3) -> dict:
    """Times the CPU restatement of the reference self-play path (games from the empty board, as
    train_alpha_zero.py:33-55 plays them, one after the other) on a bounded sample: up to `games` complete
    games inside `budget_s` seconds; `value` is 1 / the median game time, the fastest game is reported too.
    (The reference's search is deterministic, so the games are repetitions of one game: the spread is the host's.)"""
    import statistics
    import torch
    from oracle import ref_port as rp
    sd = rp.seeded_state_dict(0)
    ev = rp.make_net_evaluator(sd)
    plies_per_game = 14.0 if sims == 200 else 24.0      # length of that game when the budget cuts it short
    t_start = time.perf_counter()
    game_s, total_plies, cut = [], 0, None
    while len(game_s) < games and cut is None:
        board, player = rp.new_board(), 1
        t0 = time.perf_counter()
        plies = 0
        while not rp.is_terminal(board):
            a, _ = rp.compat_search(board, player, sims, C_PUCT, ev)
            _, player = rp.make_move(board, player, a)
            plies += 1
            if time.perf_counter() - t_start > budget_s:
                cut = (plies, time.perf_counter() - t0)
                break
        else:
            game_s.append(time.perf_counter() - t0)
            plies_per_game = float(plies)
        total_plies += plies
    dt = time.perf_counter() - t_start
    sims_per_s = total_plies * sims / dt
    if game_s:
        med = statistics.median(game_s)
        games_per_s = 1.0 / med
        sample = (f"{len(game_s)} complete game(s) of {int(plies_per_game)} plies x {sims} sims in "
                  + ", ".join(f"{g:.1f}" for g in game_s) + f" s; value = 1 / median ({med:.1f} s)")
    else:
        games_per_s = sims_per_s / (sims * max(plies_per_game, 1.0))
        sample = (f"first {cut[0]} plies x {sims} sims of one game in {cut[1]:.1f} s; games/s extrapolated to "
                  f"{plies_per_game:.1f} plies per game")
    out = {"value": games_per_s, "unit": "games/s", "sims_per_s": sims_per_s,
           "cores": torch.get_num_threads(), "kind": "port", "sample": sample}
    if game_s:
        out["best_games_per_s"] = 1.0 / min(game_s)
        out["game_seconds"] = [round(g, 2) for g in game_s]
    return out


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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games (slots) per GPU")
    ap.add_argument("--games-per-step", type=int, default=0,
                    help="finished games per GPU a step waits for (default: --games)")
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

    # NNB_BENCH_WATCHDOG_S=N: a rank that has not finished after N seconds dumps every thread's Python stack to stderr and
    # exits (rehearsals and debugging: a stalled rank otherwise hangs its launcher without a word)
    if os.environ.get("NNB_BENCH_WATCHDOG_S"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NNB_BENCH_WATCHDOG_S"]), exit=True)

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
    # Rehearsal of the multi-rank code on a box with ONE GPU (never used by a real run): every rank on cuda:0 and
    # the collectives over gloo, since RCCL refuses two ranks on one device.
    rehearsal = os.environ.get("NNB_BENCH_REHEARSAL_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
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
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=device)
            dist.barrier()
            torch.cuda.synchronize(device)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    G = args.games                                             # slots
    NG = args.games_per_step if args.games_per_step > 0 else G # games a step waits for
    PAD = max(G // 8, 64)                                      # a step ends on a 64-round boundary: a few games more finish
    extras = world == 1 and not args.no_extra_lines
    engine = Engine(max_games=G, max_sims=args.sims, device=device, precision=args.precision,
                    eval_cache_slots=2048 if extras else 0)
    engine.load_state_dict(seeded_state_dict(0))
    params = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", one_stream=args.one_stream)
    # what every rank contributes to the replay buffer per step: the step's finished records (NG + PAD rows, the
    # rows past the count repeat the last record) - fixed size, so one all-gather moves them
    local = torch.empty((NG + PAD, TRAJECTORY_BYTES), dtype=torch.uint8, device=device)
    replay = torch.empty((world * (NG + PAD), TRAJECTORY_BYTES), dtype=torch.uint8, device=device) if use_dist else local

    # start positions are inputs: resident in HBM before the timed region starts
    N_STARTS = 16384
    starts = (None, None)
    if args.opening_plies > 0:
        b, p = random_openings(engine, N_STARTS, args.opening_plies, seed=1234 + rank)
        starts = (torch.from_numpy(b.view("int64")).to(device), torch.from_numpy(p).to(device))

    state = {"started": False}

    def one_step(eng=engine, prm=params, key="started", gather=True):
        """Waits for NG more finished games of the endless stream of self-play games on the G slots."""
        first = not state.get(key, False)
        state[key] = True
        rec, stats = eng.selfplay_stream(NG, prm if first else None, restart=first, first_game_id=rank << 24,
                                         start_boards=starts[0], start_players=starts[1])
        if gather:
            n = min(rec.shape[0], NG + PAD)
            local[:n] = rec[:n]
            local[n:] = rec[n - 1]
            if use_dist:
                dist.all_gather_into_tensor(replay, local)     # RCCL: trajectories into every rank's buffer
        return stats

    def one_plain_round(eng, prm, n_games, start):
        out = torch.empty((n_games, TRAJECTORY_BYTES), dtype=torch.uint8, device=device)
        _, stats = eng.selfplay_round(n_games, prm, first_game_id=0, out=out, start_boards=start[0], start_players=start[1])
        return stats

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    # The stream is seated before the timed region: the games in flight when it starts were begun by the warm-up
    # steps, as those in flight when it ends are left to the steps that would follow (steady state).  With
    # --warmup 0 the timed region includes seating all G games at once and their common first moves.
    for w in range(args.warmup):
        one_step()
    engine.profile(True)
    fence()
    t0 = time.perf_counter()
    totals = {}
    for k in range(args.steps):
        st = one_step()
        for key, v in st.items():
            totals[key] = v if key in ("chains", "finished_total") else totals.get(key, 0) + v
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_rank0 = elapsed
    prof = engine.profile_report()
    clocks = engine.profile_clocks() if args.precision == "f32s" else {}
    engine.profile(False)

    def timed(fn, note):
        fence()
        c0 = time.perf_counter()
        st = fn()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - c0
        return {"games_per_s": st["games"] / dt, "sims_per_s": st["sims"] / dt, "seconds": dt, "games": st["games"],
                "plies_per_game": st["plies"] / max(st["games"], 1), "net_leaves": st["net_leaves"],
                "cache_hits": st["cache_hits"], "note": note}

    # Extra measurements on one GPU, each timed on its own and none part of `value`.
    extra = {}
    prof_one = None
    if extras and totals.get("chains", 1) > 1:
        # The timed region runs two tree step -> forward pass chains on two streams, so a kernel there shares the
        # chip with the other chain's kernel and its launch takes longer than its work does.  One more step with
        # all games in one chain times every kernel with the chip to itself.
        p1 = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", one_stream=True)
        one_step(engine, p1, "one_chain", gather=False)
        engine.profile(True)
        st1 = one_step(engine, p1, "one_chain", gather=False)
        prof_one = (engine.profile_report(), st1, engine.profile_clocks() if args.precision == "f32s" else {})
        engine.profile(False)
    if extras:
        extra["single_round_identical_games"] = timed(
            lambda: one_plain_round(engine, params, G, (None, None)),
            f"one plain round (nnb_selfplay_round) of {G} games from the empty board, no refill: the reference's search "
            "is deterministic, so this is one game played on every slot, all slots in step (best-case batch fill)")
        extra["single_round_distinct_games"] = timed(
            lambda: one_plain_round(engine, params, 4 * G, (starts[0], starts[1])),
            f"one plain round of {4 * G} games from the timed region's start positions on {G} slots, played until the "
            "last game has ended (the slots drain at the end of the round)")
        cparams = engine.params(args.sims, C_PUCT, evaluator="net", mode="compat", eval_cache=True)
        for _ in range(3):
            one_step(engine, cparams, "cache", gather=False)
        extra["with_eval_cache"] = timed(
            lambda: one_step(engine, cparams, "cache", gather=False),
            "one more step of the same stream with the per-game evaluation cache: a leaf whose position its game has "
            "already evaluated is not sent to the network again; same trajectories; the reference evaluates every leaf")
        azp = engine.params(args.sims, C_PUCT, evaluator="net", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=0)
        az_state = {"first": True}

        def az_step():
            first, az_state["first"] = az_state["first"], False
            _, st = engine.selfplay_stream(NG, azp if first else None, restart=first)
            return st
        for _ in range(2):
            az_step()
        extra["az_mode"] = timed(
            az_step,
            "one step of a stream of games from the empty board under the PUCT rule with the network's priors and "
            "engine-drawn Dirichlet(0.3) root noise mixed in at 0.25 (counter-based RNG, seed 0): a search the reference "
            "does not have (its expand drops the priors), held to this repository's own CPU restatement only")
        if args.precision != "f16":
            e16 = Engine(max_games=G, max_sims=args.sims, device=device, precision="f16")
            e16.load_state_dict(seeded_state_dict(0))
            p16 = e16.params(args.sims, C_PUCT, evaluator="net", mode="compat")
            for _ in range(3):
                one_step(e16, p16, "f16", gather=False)
            extra["f16_inputs"] = timed(
                lambda: one_step(e16, p16, "f16", gather=False),
                "one step of the same stream with the inputs of every product rounded to half precision (f32 accumulate): "
                "narrower than the reference's float32, so not comparable with `value`")
            e16.close()

    if extras:
        # the tree kernels without the network (the pure-tree configuration of BASELINE.json): one plain round of G games
        # at 800 simulations per move under PUCT, uniform priors, a random play-out on bitboards as the leaf's value
        et = Engine(max_games=G, max_sims=800, device=device, precision=args.precision)
        pt = et.params(800, C_PUCT, evaluator="rollout", mode="az", seed=7)
        et.selfplay_round(min(G, 256), pt)
        kept = {}

        def tree_round():
            kept["st"] = et.selfplay_round(G, pt)[1]
            return kept["st"]
        line = timed(
            tree_round,
            f"one plain round of {G} games at 800 simulations per move under PUCT with uniform priors and a uniformly "
            "random play-out on bitboards as every leaf's value (no network): the whole move of a game in one launch of "
            "the fused tree kernel, sixteen games per CU; hbm_frac = algorithmic bytes (112 x mean depth + 212 per "
            "simulation) over the round's wall time against 8 TB/s: small by design, the kernel is a chain of "
            "dependent loads per game")
        depth = kept["st"]["depth_sum"] / max(kept["st"]["sims"], 1)
        line["mean_selection_depth"] = depth
        line["algorithmic_bytes_per_sim"] = 112.0 * depth + 212.0
        line["achieved_gbs"] = line["sims_per_s"] * line["algorithmic_bytes_per_sim"] / 1e9
        line["hbm_frac"] = line["achieved_gbs"] / PEAK_HBM_GBS
        extra["tree_only_800_sims"] = line
        et.close()

    # all-gather cost alone (reported, also part of the timed region above)
    gather_ms = None
    if use_dist:
        fence()
        g0 = time.perf_counter()
        dist.all_gather_into_tensor(replay, local)
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
        if shares.get("conv4", 0.0) == 0.0 and shares.get("conv2", 0.0) > 0.0 and args.precision == "f32s":
            # the float32-faithful path runs conv0 + conv2 + conv4 as ONE launch, timed as "conv2"
            FLOP_PER_LEAF["conv2"] += FLOP_PER_LEAF.pop("conv0") + FLOP_PER_LEAF.pop("conv4")
            FLOP_PER_LEAF["conv0"] = FLOP_PER_LEAF["conv4"] = 0
            KERNEL_NAME["conv2"] = "conv0+conv2+conv4 (conv_board_split_kernel)"
        dom = max(FLOP_PER_LEAF, key=lambda k: shares.get(k, 0.0))
        avg_ms = prof[dom]["avg_ms"]
        achieved = (leaves_per_launch * FLOP_PER_LEAF[dom]) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        net_ms = sum(prof[k]["avg_ms"] for k in FLOP_PER_LEAF)
        net_tflops = (leaves_per_launch * sum(FLOP_PER_LEAF.values())) / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
        standard = G == GAMES_PER_GPU and args.sims == SIMS and args.opening_plies == OPENING_PLIES
        traffic, traffic_src = committed_traffic(args.precision, dom, leaves_per_launch) if standard else (None, None)
        tree_traffic, _ = committed_traffic(args.precision, "tree_step", leaves_per_launch) if standard else (None, None)
        plies_per_game = job["plies"] / max(games, 1)
        depth_mean = job["depth_sum"] / max(job["sims"], 1)
        tree_bytes_per_sim = 112.0 * depth_mean + 184.0
        tree_ms = prof["tree_step"]["avg_ms"]
        sims_per_launch = totals["sims"] / steps_launched
        tree_gbs = (sims_per_launch * tree_bytes_per_sim) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
        mfma_per_product = 3 if args.precision == "f32s" else 1
        # FLOPs of the matrix instructions the kernel issues per leaf: three per product on the float32-faithful path,
        # minus, in its convolutions, the (tile, tap) pairs on the zero border that are skipped (4 waves x 3456
        # 16x16x32 instructions per board + the 2->128 layer), which the algorithmic count includes
        issued_flop = {k: v * mfma_per_product for k, v in FLOP_PER_LEAF.items()}
        if args.precision == "f32s":
            issued_flop["conv2"] = 4 * 3456 * 16384 + 4 * 8 * 32768

        def issued(tflops_algorithmic):
            t = tflops_algorithmic * issued_flop[dom] / FLOP_PER_LEAF[dom]
            return {"issued_tflops": t, "issued_frac": t / PEAK_TFLOPS[args.precision],
                    "issued_flop_per_leaf": issued_flop[dom],
                    "power_capped_ceiling": {"tflops": POWER_CAPPED_MFMA_TFLOPS, "frac": t / POWER_CAPPED_MFMA_TFLOPS,
                                             "source": "profiles/r03_mfma_power_ceiling.txt: what the matrix pipes of one "
                                                       "box deliver from registers with changing operands and nothing "
                                                       "else running (1400 W cap, 1.62 GHz); `peak` is the datasheet's"}}
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
                                   "reference search rule (compat), an endless stream of self-play games (a finished "
                                   f"game's slot takes the next game); a step waits for {NG} more finished games per GPU; "
                                   "games counted = games finished inside the timed region",
                       "games_per_gpu": G, "finished_games_per_step_per_gpu": NG, "sims_per_move": args.sims, "c_puct": C_PUCT,
                       "opening_plies": args.opening_plies,
                       "parallelism": f"games sharded over {world} GPU(s), no collective inside the search; "
                                      "one all-gather of trajectories per round"},
            "stats": {"games": games, "plies": job["plies"], "sims": job["sims"], "net_leaves": job["net_leaves"],
                      "terminal_hits": job["terminal_hits"], "plies_per_game": plies_per_game,
                      "mean_selection_depth": depth_mean,
                      "leaf_batch_fill": job["net_leaves"] / max(job["steps"] * G, 1),
                      "lockstep_rounds": job["steps"], "launch_chains_per_round": chains,
                      "allgather_ms": gather_ms,
                      "allgather_bytes": world * (NG + PAD) * TRAJECTORY_BYTES if use_dist else 0},
            "kernel_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()},
            "roofline": {"kernel": KERNEL_NAME.get(dom, dom), "bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.precision],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": traffic,
                         "traffic_source": traffic_src,
                         "flop_per_leaf": FLOP_PER_LEAF[dom], "leaves_per_launch": leaves_per_launch,
                         "avg_launch_ms": avg_ms, "whole_net_tflops": net_tflops,
                         "mfma_per_product": mfma_per_product, **issued(achieved),
                         "concurrent_chains": chains,
                         "shader_clock_mhz": {k: round(v) for k, v in clocks.items()}, "peak_assumes_mhz": 2400,
                         "region_net_tflops": totals["net_leaves"] * sum(FLOP_PER_LEAF.values()) / (elapsed_rank0 * 1e12),
                         "region_frac": totals["net_leaves"] * sum(FLOP_PER_LEAF.values()) / (elapsed_rank0 * 1e12)
                                        / PEAK_TFLOPS[args.precision]},
            "roofline_tree": {"kernel": "tree_step", "bound": "hbm", "achieved": tree_gbs, "peak": PEAK_HBM_GBS,
                              "unit": "GB/s", "frac": tree_gbs / PEAK_HBM_GBS, "traffic": tree_traffic,
                              "bytes_per_sim": tree_bytes_per_sim, "sims_per_launch": sims_per_launch,
                              "avg_launch_ms": tree_ms},
        }
        if prof_one is not None:
            out["roofline_timed_region"] = out["roofline"]
            out["roofline_timed_region"]["note"] = (
                "launch durations by HIP events on each chain's own stream inside the timed region: a launch there shares "
                "the chip with the other chain's kernels, so its duration overstates the time its work takes; region_frac "
                "is the network's algorithmic FLOPs of the whole timed region over the region's wall time")
            p1, st1, clk1 = prof_one
            lpl = st1["net_leaves"] / max(st1["steps"], 1)
            ms = p1[dom]["avg_ms"]
            ach = lpl * FLOP_PER_LEAF[dom] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            tr, src = committed_traffic(args.precision, dom, lpl) if standard else (None, None)
            out["roofline"] = {
                "kernel": KERNEL_NAME.get(dom, dom), "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                "frac": ach / PEAK_TFLOPS[args.precision], "traffic": tr, "traffic_source": src,
                "leaves_per_launch": lpl, "avg_launch_ms": ms,
                **issued(ach),
                "flop_per_leaf": FLOP_PER_LEAF[dom], "mfma_per_product": mfma_per_product,
                "shader_clock_mhz": {k: round(v) for k, v in clk1.items()}, "peak_assumes_mhz": 2400,
                "kernel_ms": {k: round(v["avg_ms"], 4) for k, v in p1.items()},
                "measured": "OUTSIDE the timed region: HIP events on the launch stream over one more step of the same stream of games, run right "
                            "after the timed region with all games in ONE chain, so that every kernel has the chip to "
                            "itself (the timed region runs two chains on two streams: see roofline_timed_region; "
                            "`bench.py --one-stream` times the whole run in one chain, and profiles/*_one_chain_* is "
                            "its rocprofv3 summary)"}
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims)
            out["cpu_baseline"]["config1_100_sims"] = cpu_baseline(100, budget_s=15.0, games=1)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()

"""Tree-only self-play (random rollouts, no network) against the simulation count: how many games share a CU is what
the fused tree kernel's throughput follows.  Usage (GPU box): [NNB_LIB=variant.so] python tools/tree_occupancy.py [sims ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from alphazero_amd.engine import Engine  # noqa: E402

G = 4096
for sims in [int(x) for x in sys.argv[1:]] or [100, 200, 300, 400, 800]:
    eng = Engine(max_games=G, max_sims=sims)
    for mode in ("az", "compat"):
        p = eng.params(sims, 1.4, evaluator="rollout", mode=mode, seed=0)
        eng.selfplay_round(256, p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        traj, st = eng.selfplay_round(G, p)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mode} S={sims}: {st['sims'] / dt / 1e6:.1f} M sims/s, {st['games'] / dt:.0f} games/s, "
              f"depth {st['depth_sum'] / max(st['sims'], 1):.2f}", flush=True)
    eng.close()

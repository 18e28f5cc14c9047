"""Times each stage of the network forward pass on the GPU (device events inside libnnb)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.engine import Engine          # noqa: E402
from alphazero_amd.network import seeded_state_dict  # noqa: E402

FLOP = {"conv0": 2 * 128 * 56 * 32, "conv2": 2 * 128 * 72 * 2048, "conv4": 2 * 128 * 90 * 2048,
        "fc": 2 * 11520 * 1024, "heads": 2 * 1024 * 8}

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--precision", default="f16")
args = ap.parse_args()
eng = Engine(max_games=args.n, max_sims=8, precision=args.precision)
eng.load_state_dict(seeded_state_dict(0))
rng = np.random.RandomState(0)
cells = rng.randint(0, 3, size=(args.n, 42))
side = np.zeros((args.n, 2), dtype=np.uint64)
for c in range(42):
    bit = np.uint64(1 << ((c % 7) * 7 + c // 7))
    side[:, 0] |= np.where(cells[:, c] == 1, bit, np.uint64(0))
    side[:, 1] |= np.where(cells[:, c] == 2, bit, np.uint64(0))
side_t = torch.from_numpy(side.view(np.int64)).cuda()
for _ in range(5):
    eng.net_forward(side_t)
torch.cuda.synchronize()
eng.profile(True)
for _ in range(args.iters):
    eng.net_forward(side_t)
    torch.cuda.synchronize()
    eng.lib.nnb_search  # keep flake quiet
# profile samples are drained by search / self-play calls; force a drain through a tiny search
eng.search(np.zeros((1, 2), dtype=np.uint64), np.array([1], dtype=np.int8), eng.params(2, 1.4, evaluator="const"))
rep = eng.profile_report()
tot = 0.0
for k in ("conv0", "conv2", "conv4", "fc", "heads"):
    ms = rep[k]["avg_ms"]
    tot += ms
    print(f"{k:6s} {ms*1e3:9.1f} us  {args.n * FLOP[k] / (ms * 1e-3) / 1e12 if ms else 0:8.1f} TFLOP/s  (samples {rep[k]['samples']})")
print(f"total  {tot*1e3:9.1f} us  {args.n * sum(FLOP.values()) / (tot * 1e-3) / 1e12:8.1f} TFLOP/s")

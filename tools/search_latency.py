import time, torch, numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.engine import Engine
from alphazero_amd.network import seeded_state_dict
# usage: search_latency.py [precision[,precision] [roots ...]]
precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ("f32s", "f16")
roots = [int(a) for a in sys.argv[2:]] or (1, 8, 64, 128)
for prec in precs:
    for n in roots:
        eng = Engine(max_games=n, max_sims=800, precision=prec)
        eng.load_state_dict(seeded_state_dict(0))
        b = np.zeros((n, 2), dtype=np.uint64); p = np.ones(n, dtype=np.int8)
        prm = eng.params(800, 1.4, evaluator="net")
        eng.search(b, p, prm)
        eng.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.search(b, p, prm)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(prec, n, "roots x 800 sims: %.1f ms" % (dt * 1e3), {k: round(v["avg_ms"] * 1e3, 1) for k, v in eng.profile_report().items() if v["avg_ms"] > 0},
              {k: round(v) for k, v in eng.profile_clocks().items()} if prec == "f32s" else "", flush=True)
        eng.close()

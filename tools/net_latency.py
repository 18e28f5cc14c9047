"""Forward-pass time of the float32-faithful network at middling batch sizes (profile classes of one call)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.engine import Engine
from alphazero_amd.network import seeded_state_dict
eng = Engine(max_games=4096, max_sims=8)
eng.load_state_dict(seeded_state_dict(0))
sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048]
for n in sizes:
    side = np.zeros((n, 2), dtype=np.uint64)
    side_t = torch.from_numpy(side.view("int64")).to(eng.device)
    eng.net_forward(side_t)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        eng.net_forward(side_t)
    b.record()
    torch.cuda.synchronize()
    print(n, "rows: %.1f us per forward pass" % (a.elapsed_time(b) / 20 * 1e3), flush=True)
eng.close()

"""How many tree step -> forward pass chains should share the chip?  The engine runs one or two; this experiment puts
several engines side by side in one process (one host thread and one stream each), so that 4096 slots are played as
1, 2, 3 or 4 chains, and reports the aggregate self-play rate.  Usage: python tools/chains_experiment.py [steps]"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.engine import Engine
from alphazero_amd.network import seeded_state_dict
from alphazero_amd.selfplay import random_openings

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
SIMS, TOTAL = 200, 4096
dev = torch.device("cuda", 0)


def run(layout):
    """layout: list of (slots, chains) per engine"""
    engines, streams, prms, starts = [], [], [], []
    for i, (slots, chains) in enumerate(layout):
        e = Engine(max_games=slots, max_sims=SIMS, device=dev, precision="f32s")
        e.load_state_dict(seeded_state_dict(0))
        b, p = random_openings(e, 8192, 4, seed=77 + i)
        starts.append((torch.from_numpy(b.view("int64")).to(dev), torch.from_numpy(p).to(dev)))
        prms.append(e.params(SIMS, 1.4, evaluator="net", mode="compat", one_stream=chains == 1, two_streams=chains == 2))
        engines.append(e); streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    bar = threading.Barrier(len(layout))
    out = [None] * len(layout)

    def worker(i):
        e, slots = engines[i], layout[i][0]
        with torch.cuda.stream(streams[i]):
            e.selfplay_stream(slots, prms[i], restart=True, first_game_id=i << 24, start_boards=starts[i][0], start_players=starts[i][1])
            e.selfplay_stream(slots, None, restart=False, first_game_id=i << 24, start_boards=starts[i][0], start_players=starts[i][1])
            streams[i].synchronize()
            bar.wait()
            t0 = time.perf_counter(); games = sims = 0
            for _ in range(STEPS):
                _, st = e.selfplay_stream(slots, None, restart=False, first_game_id=i << 24, start_boards=starts[i][0], start_players=starts[i][1])
                games += st["games"]; sims += st["sims"]
            streams[i].synchronize()
            out[i] = (games, sims, time.perf_counter() - t0)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(layout))]
    for t in th: t.start()
    for t in th: t.join()
    for e in engines: e.close()
    # engines finish at slightly different times: rate = sum of the engines' own rates
    return sum(g / dt for g, s, dt in out), sum(s / dt for g, s, dt in out)


layouts = {
    "1 chain  (one engine, 4096 slots, one chain)": [(4096, 1)],
    "2 chains (one engine, 4096 slots, two chains: the default)": [(4096, 2)],
    "2 chains (two engines x 2048 slots, one chain each)": [(2048, 1), (2048, 1)],
    "3 chains (2736 slots in two chains + 1360 slots in one)": [(2736, 2), (1360, 1)],
    "3 chains (three engines x 1360/1360/1376 slots, one chain each)": [(1360, 1), (1360, 1), (1376, 1)],
    "4 chains (two engines x 2048 slots, two chains each)": [(2048, 2), (2048, 2)],
}
for name, layout in layouts.items():
    g, s = run(layout)
    print(f"{name}: {g:7.0f} games/s  {s / 1e6:5.2f} M sims/s", flush=True)

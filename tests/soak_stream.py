"""Not collected by pytest (a long run, started by hand): soak run of the benchmark's workload: a long stream of self-play games (4096 slots, 200 sims/move, network,
float32-faithful, two chains), every finished record replayed against the rule oracle, duplicates of one start
position compared byte for byte, and a sample re-played as a plain round.  Usage: python tests/soak_stream.py [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd.engine import Engine, trajectories_to_numpy
from alphazero_amd.network import seeded_state_dict
from alphazero_amd.selfplay import random_openings
from oracle import c_oracle as co

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G, S, NST = 4096, 200, 16384
eng = Engine(max_games=G, max_sims=S)
eng.load_state_dict(seeded_state_dict(0))
p = eng.params(S, 1.4, evaluator="net", mode="compat")
sb, sp = random_openings(eng, NST, 4, seed=1234)
recs = []
t0 = time.perf_counter()
for i in range(steps):
    r, st = eng.selfplay_stream(G, p if i == 0 else None, restart=(i == 0), start_boards=sb, start_players=sp)
    recs.append(trajectories_to_numpy(r).copy())
    print(f"step {i}: {st['games']} games, {st['sims']} sims, {time.perf_counter() - t0:.1f} s", flush=True)
rec = np.concatenate(recs)
ids = rec["game_id"]
assert len(set(ids.tolist())) == len(ids), "a game was listed twice"
print(len(rec), "records; replaying", flush=True)
lib = co.lib()
first = {}
for n, r in enumerate(rec):
    j = int(r["game_id"]) % NST
    b1, b2, player = int(sb[j][0]), int(sb[j][1]), int(sp[j])
    L = int(r["length"])
    assert r["result"] != 2 and 1 <= L <= 42
    for k in range(L):
        ply = r["plies"][k]
        assert (int(ply["colour_boards"][0]), int(ply["colour_boards"][1])) == (b1, b2) and int(ply["player"]) == player
        v = ply["visits"].astype(np.int64)
        a = int(ply["action"])
        assert int(v.sum()) == S - 1 and v[a] == v.max() and a == int(np.argmax(v))
        ok, b1, b2 = co.drop(b1, b2, player, a)
        assert ok
        player = 3 - player
    assert (int(r["final_boards"][0]), int(r["final_boards"][1])) == (b1, b2) and lib.orc_is_terminal(b1, b2)
    assert int(r["result"]) == lib.orc_result(b1, b2)
    key = (int(sb[j][0]), int(sb[j][1]), int(sp[j]))
    if key in first:
        assert r["plies"].tobytes() == rec[first[key]]["plies"].tobytes(), (n, first[key])
    else:
        first[key] = n
    if n % 20000 == 0:
        print("replayed", n, flush=True)
print("distinct start positions:", len(first), flush=True)
# a sample of the games again as one plain round (one chain of big kernels for most of it, small kernels at its end)
sel = np.sort(np.random.RandomState(0).choice(len(rec), 3000, replace=False))
j = rec["game_id"][sel] % NST
traj, _ = eng.selfplay_round(len(sel), eng.params(S, 1.4, evaluator="net", mode="compat", one_stream=True),
                             start_boards=sb[j], start_players=sp[j])
ref = trajectories_to_numpy(traj)
assert all(ref[i]["plies"].tobytes() == rec[sel[i]]["plies"].tobytes() for i in range(len(sel)))
print("soak ok:", len(rec), "games", flush=True)

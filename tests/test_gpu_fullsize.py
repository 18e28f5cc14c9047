"""The benchmark configurations at their full sizes, checked through properties that do not need
the CPU search to play them all: every recorded game must replay under the rules, the visit
counts of a move must add up to the simulation budget, identical games must come out identical
wherever they sit in the batch, and a sample of the games is compared with the CPU search move
for move."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


@pytest.fixture(scope="module")
def oracle():
    from oracle import c_oracle as co
    from oracle import ref_port as rp
    return co, rp


def _replay_and_check(co, rec, sims, start_players=None, sample=None, start_boards=None):
    """Replays every record with the rule oracle; returns the set of distinct move sequences."""
    games = set()
    idx = range(len(rec)) if sample is None else sample
    for i in idx:
        r = rec[i]
        n = int(r["length"])
        b1 = b2 = 0
        if start_boards is not None:
            b1, b2 = int(start_boards[i][0]), int(start_boards[i][1])
        player = 1 if start_players is None else int(start_players[i])
        assert (7 if start_boards is None else 1) <= n <= 42
        for k in range(n):
            ply = r["plies"][k]
            assert (int(ply["colour_boards"][0]), int(ply["colour_boards"][1])) == (b1, b2)
            assert int(ply["player"]) == player
            assert not co.lib().orc_is_terminal(b1, b2)
            legal = co.legal_mask(b1, b2)
            v = ply["visits"].astype(np.int64)
            assert int(v.sum()) == sims - 1                     # every simulation but the first descends a root child
            assert all(v[c] == 0 for c in range(7) if not (legal >> c) & 1)
            a = int(ply["action"])
            assert v[a] == v.max() and a == int(np.argmax(v))   # most visited, lowest column on ties
            ok, b1, b2 = co.drop(b1, b2, player, a)
            assert ok
            player = 3 - player
        assert (int(r["final_boards"][0]), int(r["final_boards"][1])) == (b1, b2)
        assert co.lib().orc_is_terminal(b1, b2)
        assert int(r["result"]) == co.lib().orc_result(b1, b2)
        assert not r["plies"][n:].tobytes().strip(b"\0")        # unused plies are zero
        games.add(r["plies"]["action"][:n].tobytes())
    return games


def test_config2_4096_games_200_sims_network(gpu, oracle):
    """4096 concurrent games x 200 sims/move with the seed-0 network (the benchmark workload).  The
    reference search is deterministic, so all games from the empty board are one and the same game:
    4096 identical records prove that a leaf's evaluation does not depend on its place in the batch."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, rp = oracle
    eng = Engine(max_games=4096, max_sims=200)
    try:
        eng.load_state_dict(rp.seeded_state_dict(0))
        traj, stats = eng.selfplay_round(4096, eng.params(200, 1.4, evaluator="net"))
        rec = trajectories_to_numpy(traj)
    finally:
        eng.close()
    assert (rec["game_id"] == np.arange(4096)).all()
    first = rec[0]
    n = int(first["length"])
    assert (rec["length"] == n).all() and (rec["result"] == first["result"]).all()
    assert all(r["plies"].tobytes() == first["plies"].tobytes() for r in rec)
    _replay_and_check(co, rec, 200, sample=[0, 4095])
    assert stats["games"] == 4096 and stats["plies"] == 4096 * n and stats["sims"] == 4096 * n * 200
    assert stats["net_leaves"] + stats["terminal_hits"] == stats["sims"]
    # the whole game, move for move and visit vector for visit vector, equals the CPU search (the reference's
    # rule, oracle/c4_oracle.c) driven by the CPU float32 network (train_alpha_zero.py:41-45)
    from alphazero_amd.game import bits_to_board
    net = rp.make_net_evaluator(rp.seeded_state_dict(0))
    b1 = b2 = 0
    player = 1
    for k in range(n):
        a, v, _ = co.compat_search(b1, b2, player, 200, 1.4, lambda me, opp: net(bits_to_board(me, opp), 1))
        assert v == first["plies"]["visits"][k].tolist() and a == int(first["plies"]["action"][k]), k
        _, b1, b2 = co.drop(b1, b2, player, a)
        player = 3 - player
    assert co.lib().orc_is_terminal(b1, b2)


def test_benchmark_stream_4096_slots_two_chains_equals_plain_round_and_cpu_search(gpu, oracle):
    """The workload bench.py times, at its size: 4096 slots, 200 sims/move, the seed-0 network in float32-faithful
    arithmetic, games from 4 random opening plies, an endless stream on the default two chains.  Two steps of the
    stream; every record must equal, byte for byte, the record a plain round writes for the same game id and start
    position, and the first plies of four sampled games must equal the CPU search (moves and visit vectors)."""
    import torch
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.game import bits_to_board
    from alphazero_amd.selfplay import random_openings
    co, rp = oracle
    slots, sims, n_starts = 4096, 200, 16384
    eng = Engine(max_games=slots, max_sims=sims)
    assert eng.precision == "f32s"
    try:
        eng.load_state_dict(rp.seeded_state_dict(0))
        sb, sp = random_openings(eng, n_starts, 4, seed=1234)
        prm = eng.params(sims, 1.4, evaluator="net", mode="compat")
        recs = []
        for i in range(2):
            r, st = eng.selfplay_stream(slots, prm if i == 0 else None, restart=(i == 0), start_boards=sb, start_players=sp)
            assert st["chains"] == 2 and st["games"] >= slots
            recs.append(trajectories_to_numpy(r).copy())
        rec = np.concatenate(recs)
        ids = rec["game_id"]
        assert len(set(ids.tolist())) == len(ids) and (rec["result"] != 2).all()
        n = int(ids.max()) + 1
        assert n <= n_starts
        traj, st = eng.selfplay_round(n, prm, start_boards=sb[:n], start_players=sp[:n])
        ref = trajectories_to_numpy(traj)
        assert rec.tobytes() == ref[ids].tobytes()
    finally:
        eng.close()
    _replay_and_check(co, rec, sims, start_players=sp[ids], sample=range(0, len(rec), 97), start_boards=sb[ids])
    net = rp.make_net_evaluator(rp.seeded_state_dict(0))
    for i in (0, len(rec) // 3, 2 * len(rec) // 3, len(rec) - 1):
        r = rec[i]
        b1, b2, player = int(sb[ids[i]][0]), int(sb[ids[i]][1]), int(sp[ids[i]])
        for k in range(min(3, int(r["length"]))):
            a, v, _ = co.compat_search(b1, b2, player, sims, 1.4, lambda me, opp: net(bits_to_board(me, opp), 1))
            assert v == r["plies"]["visits"][k].tolist() and a == int(r["plies"]["action"][k]), (i, k)
            _, b1, b2 = co.drop(b1, b2, player, a)
            player = 3 - player


def test_config4_tree_only_800_sims_rollouts(gpu, oracle):
    """4096 games x 800 sims/move, PUCT with uniform priors, Dirichlet noise and random rollouts."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, _ = oracle
    seed, first_id = 2024, 100000
    eng = Engine(max_games=4096, max_sims=800)
    try:
        prm = eng.params(800, 1.4, evaluator="rollout", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=seed)
        traj, stats = eng.selfplay_round(4096, prm, first_game_id=first_id)
        rec = trajectories_to_numpy(traj)
        games = _replay_and_check(co, rec, 800)
        assert len(games) > 3500                                # noise and rollouts differ per game
        assert stats["sims"] == stats["plies"] * 800
        # three of the games against the CPU search, fed the noise the engine reports for them
        for i in (0, 1234, 4095):
            r = rec[i]
            b1 = b2 = 0
            player = 1
            for k in range(int(r["length"])):
                legal = co.legal_mask(b1, b2)
                noise = eng.dirichlet_noise(seed, [first_id + i], [k], [legal], 0.3)[0]
                a, v, _ = co.az_search(b1, b2, player, 800, 1.4, "rollout", root_noise=noise, eps=0.25, seed=seed,
                                       game_id=first_id + i, move_index=k)
                assert v == r["plies"]["visits"][k].tolist() and a == int(r["plies"]["action"][k]), (i, k)
                _, b1, b2 = co.drop(b1, b2, player, a)
                player = 3 - player
    finally:
        eng.close()


def test_config5_arena_10000_games_400_sims(gpu, oracle):
    """10 000 arena games, 400 sims/move, two evaluators, starter alternates, 4096 slots refilled.
    With fixed evaluators the games of one starter are all the same game; both are checked
    against the CPU search."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, rp = oracle
    mix = (0xD6E8FEB86659FD93) & ((1 << 64) - 1)
    evs = [lambda me, opp: rp.hash_value_q8(me, opp), lambda me, opp: rp.hash_value_q8(me ^ mix, opp)]
    n = 10000
    starters = (1 + np.arange(n) % 2).astype(np.int8)
    eng = Engine(max_games=4096, max_sims=400, n_weight_sets=2)
    try:
        prm = eng.params(400, 1.4, evaluator="hash_q8", weight_set=0, weight_set_colour2=1)
        traj, stats = eng.selfplay_round(n, prm, start_boards=np.zeros((n, 2), dtype=np.uint64), start_players=starters)
        rec = trajectories_to_numpy(traj)
    finally:
        eng.close()
    assert stats["games"] == n
    for s in (1, 2):
        b1 = b2 = 0
        player = s
        moves, visits = [], []
        while not co.lib().orc_is_terminal(b1, b2):
            a, v, _ = co.compat_search(b1, b2, player, 400, 1.4, evs[player - 1])
            moves.append(a); visits.append(v)
            _, b1, b2 = co.drop(b1, b2, player, a)
            player = 3 - player
        mine = rec[s - 1::2]
        assert (mine["length"] == len(moves)).all()
        assert all(r["plies"]["action"][:len(moves)].tolist() == moves for r in mine[:50])
        assert all(r["plies"].tobytes() == mine[0]["plies"].tobytes() for r in mine)
        assert mine[0]["plies"]["visits"][:len(moves)].tolist() == visits
        assert (mine["result"] == co.lib().orc_result(b1, b2)).all()


def test_config5_arena_10000_games_400_sims_two_networks(gpu, oracle):
    """The arena at full size with two real networks (seed-0 weights search for colour 1, seed-1 weights for
    colour 2, run_tournament.py:12-42) in float32-faithful arithmetic: 10 000 games, 400 sims/move, 4096 slots
    refilled, starters alternate, distinct start positions.  Every 7th record is replayed against the rules;
    games with the same start position and starter must be identical; and the first plies of one game per
    starter are searched on the CPU with the reference's rule and the mover's network."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.game import bits_to_board
    from alphazero_amd.selfplay import random_openings
    co, rp = oracle
    n, sims = 10000, 400
    sds = [rp.seeded_state_dict(0), rp.seeded_state_dict(1)]
    starters = (1 + np.arange(n) % 2).astype(np.int8)
    eng = Engine(max_games=4096, max_sims=sims, n_weight_sets=2)
    assert eng.precision == "f32s"
    try:
        eng.load_state_dict(sds[0], 0)
        eng.load_state_dict(sds[1], 1)
        sb, _ = random_openings(eng, n, 4, seed=77)
        prm = eng.params(sims, 1.4, evaluator="net", weight_set=0, weight_set_colour2=1)
        traj, stats = eng.selfplay_round(n, prm, start_boards=sb, start_players=starters)
        rec = trajectories_to_numpy(traj).copy()
    finally:
        eng.close()
    assert stats["games"] == n and stats["sims"] == stats["plies"] * sims
    assert (rec["result"] != 2).all() and (rec["game_id"] == np.arange(n)).all()
    games = _replay_and_check(co, rec, sims, start_players=starters, sample=range(0, n, 7), start_boards=sb)
    assert len(games) > 300
    seen = {}
    for i in range(n):
        key = (int(sb[i][0]), int(sb[i][1]), int(starters[i]))
        j = seen.setdefault(key, i)
        assert rec[i]["plies"].tobytes() == rec[j]["plies"].tobytes(), (i, j)
    assert len(seen) > 500
    nets = [rp.make_net_evaluator(sd) for sd in sds]
    evs = [lambda me, opp, f=f: f(bits_to_board(me, opp), 1) for f in nets]
    for i in (0, 1):                                             # one game per starter, its first two plies
        r = rec[i]
        b1, b2, player = int(sb[i][0]), int(sb[i][1]), int(starters[i])
        for k in range(min(2, int(r["length"]))):
            a, v, _ = co.compat_search(b1, b2, player, sims, 1.4, evs[player - 1])
            assert v == r["plies"]["visits"][k].tolist() and a == int(r["plies"]["action"][k]), (i, k)
            _, b1, b2 = co.drop(b1, b2, player, a)
            player = 3 - player


def test_config2_puct_with_network_is_reproducible_and_slot_independent(gpu, oracle):
    """4096 slots x 200 sims/move under the PUCT rule with engine-drawn root noise and the network:
    every block of every launch races only for batch slots, never for data, so two runs give the
    same bytes, and 6000 games on 4096 slots contain the same first 4096 games."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, rp = oracle
    eng = Engine(max_games=4096, max_sims=200)
    try:
        eng.load_state_dict(rp.seeded_state_dict(0))
        prm = eng.params(200, 1.4, evaluator="net", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=17)
        a = trajectories_to_numpy(eng.selfplay_round(4096, prm)[0]).copy()
        b = trajectories_to_numpy(eng.selfplay_round(4096, prm)[0]).copy()
        c = trajectories_to_numpy(eng.selfplay_round(6000, prm)[0]).copy()
    finally:
        eng.close()
    assert a.tobytes() == b.tobytes()
    assert c[:4096].tobytes() == a.tobytes()
    games = _replay_and_check(co, c, 200, sample=range(0, 6000, 7))
    assert len(games) > 800


def test_engine_sizes_from_one_game_to_more_than_one_kernel_group(gpu, oracle):
    """1, 3 and 5000 slots (5000 rows span two groups of the linear layer's grid): the same
    deterministic game comes out everywhere."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    _, rp = oracle
    sd = rp.seeded_state_dict(0)
    seen = set()
    for slots in (1, 3, 5000):
        eng = Engine(max_games=slots, max_sims=64)
        try:
            eng.load_state_dict(sd)
            traj, stats = eng.selfplay_round(slots, eng.params(64, 1.4, evaluator="net"))
            rec = trajectories_to_numpy(traj)
        finally:
            eng.close()
        assert all(r["plies"].tobytes() == rec[0]["plies"].tobytes() for r in rec)
        assert stats["sims"] == slots * int(rec[0]["length"]) * 64
        seen.add(rec[0]["plies"].tobytes())
    assert len(seen) == 1

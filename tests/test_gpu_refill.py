"""Self-play rounds longer than the engine's slot count: a slot whose game ends takes the next
unplayed game (the batched form of the sequential game loop in train_alpha_zero.py:37-55).
What a game plays must depend on its id and start position only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import c_oracle as co
    from oracle import ref_port as rp
    return co, rp


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


def _positions(co, n, seed, max_plies):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        b1 = b2 = 0
        player = 1
        for _ in range(int(rng.randint(0, max_plies + 1))):
            m = co.legal_mask(b1, b2)
            cols = [c for c in range(7) if (m >> c) & 1]
            _, b1, b2 = co.drop(b1, b2, player, int(rng.choice(cols)))
            player = 3 - player
            if co.lib().orc_is_terminal(b1, b2):
                break
        out.append((b1, b2, player))          # finished start positions stay in: they are retired unplayed
    return out


@pytest.mark.parametrize("lockstep", [False, True])
def test_refill_fixed_evaluator_vs_oracle(gpu, oracle, lockstep):
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, _ = oracle
    n, slots, sims = 300, 37, 60
    pos = _positions(co, n, seed=5, max_plies=30)
    assert any(co.lib().orc_is_terminal(p[0], p[1]) for p in pos)
    eng = Engine(max_games=slots, max_sims=sims)
    try:
        traj, stats = eng.selfplay_round(n, eng.params(sims, 1.4, evaluator="hash_q8", lockstep=lockstep), first_game_id=500,
                                         start_boards=np.array([(p[0], p[1]) for p in pos], dtype=np.uint64),
                                         start_players=np.array([p[2] for p in pos], dtype=np.int8))
        rec = trajectories_to_numpy(traj)
    finally:
        eng.close()
    plies = 0
    for i, (b1, b2, player) in enumerate(pos):
        moves, visits, result, final, _ = co.selfplay_game(sims, 1.4, "hash_q8", b1, b2, player)
        r = rec[i]
        assert r["game_id"] == 500 + i
        assert r["length"] == len(moves), i
        assert r["plies"]["action"][:len(moves)].tolist() == moves
        assert r["plies"]["visits"][:len(moves)].tolist() == visits
        assert r["result"] == (2 if result is None else result)
        assert (int(r["final_boards"][0]), int(r["final_boards"][1])) == final
        plies += len(moves)
    assert stats["games"] == n and stats["plies"] == plies and stats["sims"] == plies * sims


def test_refill_az_noise_and_rollouts_independent_of_slot_count(gpu):
    """PUCT + engine-drawn Dirichlet noise + random rollouts: every draw is addressed by
    (seed, game id, move index, simulation), so 200 games on 24 slots equal 200 games on 256."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    out = []
    for slots in (24, 256):
        eng = Engine(max_games=slots, max_sims=64)
        try:
            p = eng.params(64, 1.4, evaluator="rollout", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=99)
            traj, stats = eng.selfplay_round(200, p, first_game_id=7)
            out.append(trajectories_to_numpy(traj).copy())
            assert stats["games"] == 200
        finally:
            eng.close()
    assert out[0].tobytes() == out[1].tobytes()
    assert len({r["plies"]["action"][:r["length"]].tobytes() for r in out[0]}) > 100      # games do differ


@pytest.mark.parametrize("precision", ["f16", "f32"])
def test_refill_network_games_independent_of_slot_count(gpu, oracle, precision):
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    _, rp = oracle
    sd = rp.seeded_state_dict(0)
    out = []
    for slots in (16, 128):
        eng = Engine(max_games=slots, max_sims=32, precision=precision)
        try:
            eng.load_state_dict(sd)
            p = eng.params(32, 1.4, evaluator="net", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=3)
            traj, stats = eng.selfplay_round(100, p)
            out.append(trajectories_to_numpy(traj).copy())
        finally:
            eng.close()
    assert out[0].tobytes() == out[1].tobytes()


def test_hostile_start_positions_are_sanitised(gpu):
    """Arbitrary 64-bit patterns and mover bytes must not take the kernels out of bounds: the engine
    keeps the 42 playable bits, gives a doubly owned cell to colour 1 and reads any mover but 2 as 1."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    rng = np.random.RandomState(3)
    n = 200
    boards = rng.randint(0, 2 ** 63, size=(n, 2), dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    players = rng.randint(-128, 128, size=n).astype(np.int8)
    eng = Engine(max_games=64, max_sims=32, n_weight_sets=2)
    try:
        for kw in ({"evaluator": "hash_q8"}, {"evaluator": "hash_q8", "lockstep": True, "weight_set_colour2": 1},
                   {"evaluator": "rollout", "mode": "az", "dirichlet_alpha": 0.3, "dirichlet_eps": 0.25}):
            traj, stats = eng.selfplay_round(n, eng.params(32, 1.4, **kw), start_boards=boards, start_players=players)
            rec = trajectories_to_numpy(traj)
            assert stats["games"] == n
            mask = np.uint64(0xFDFBF7EFDFBF)
            assert ((rec["final_boards"] & ~mask) == 0).all()
            assert ((rec["final_boards"][:, 0] & rec["final_boards"][:, 1]) == 0).all()
            assert np.isin(rec["result"], (-1, 0, 1, 2)).all()      # 2: no legal move yet not finished (stones in mid-air)
            v, a, _, _ = eng.search(boards[:64], players[:64], eng.params(32, 1.4, **kw))
            assert ((a >= -1) & (a < 7)).all()
    finally:
        eng.close()


@pytest.mark.parametrize("mode,evaluator", [("compat", "hash_q8"), ("az", "hash_f32"), ("az", "net"), ("compat", "net")])
def test_games_without_a_common_move_boundary_equal_synchronised_moves(gpu, oracle, mode, evaluator):
    """By default every game ends its move and starts the next one as soon as its own simulations
    are done; with sync_moves all games wait for each other.  The games must not depend on that."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, rp = oracle
    eng = Engine(max_games=40, max_sims=48, n_weight_sets=2)
    try:
        if evaluator == "net":
            eng.load_state_dict(rp.seeded_state_dict(0), 0)
            eng.load_state_dict(rp.seeded_state_dict(1), 1)
        n = 130
        pos = _positions(co, n, seed=11, max_plies=20)
        boards = np.array([(p[0], p[1]) for p in pos], dtype=np.uint64)
        players = np.array([p[2] for p in pos], dtype=np.int8)
        extra = dict(dirichlet_alpha=0.5, dirichlet_eps=0.25, seed=77) if mode == "az" else {}
        out = []
        for sync in (False, True):
            prm = eng.params(48, 1.4, evaluator=evaluator, mode=mode, lockstep=True, sync_moves=sync, weight_set=0,
                             weight_set_colour2=1, **extra)
            traj, stats = eng.selfplay_round(n, prm, first_game_id=9, start_boards=boards, start_players=players)
            out.append((trajectories_to_numpy(traj).copy(), stats))
        assert out[0][0].tobytes() == out[1][0].tobytes()
        for k in ("games", "plies", "sims", "net_leaves", "terminal_hits", "depth_sum"):
            assert out[0][1][k] == out[1][1][k], k
        assert out[0][1]["steps"] < out[1][1]["steps"]        # fewer lock-step rounds for the same games
    finally:
        eng.close()


def test_random_openings_are_legal_positions(gpu, oracle):
    from alphazero_amd.engine import Engine
    from alphazero_amd.selfplay import random_openings
    co, _ = oracle
    eng = Engine(max_games=8, max_sims=8)
    try:
        for plies in (0, 3, 6):
            boards, movers = random_openings(eng, 300, plies, seed=4)
            again, _ = random_openings(eng, 300, plies, seed=4)
            assert (boards == again).all()
            assert (movers == 1 + plies % 2).all()
            for b1, b2 in boards[:100]:
                b1, b2 = int(b1), int(b2)
                assert b1 & b2 == 0 and bin(b1).count("1") == (plies + 1) // 2 and bin(b2).count("1") == plies // 2
                assert not co.lib().orc_is_terminal(b1, b2)
                for c in range(7):                      # no stone in mid-air
                    col = ((b1 | b2) >> (7 * c)) & 0x3F
                    assert col & (col + 1) == 0
            if plies:
                assert len({(int(a), int(b)) for a, b in boards}) > 20
    finally:
        eng.close()


@pytest.mark.parametrize("mode,evaluator,precision", [("compat", "hash_f32", "f16"), ("az", "hash_q8", "f16"),
                                                      ("compat", "net", "f32s"), ("az", "net", "f16")])
def test_two_halves_on_two_streams_play_the_same_games(gpu, oracle, mode, evaluator, precision):
    """A round with at least 1024 slots runs as two halves on two unsynchronised streams (one half's tree step
    under the other half's forward pass); one_stream keeps the single chain.  Same records byte for byte,
    with refill (more games than slots) and with games of very different lengths."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, rp = oracle
    slots, n, sims = 1100, 1500, 24
    pos = _positions(co, n, seed=11, max_plies=34)
    boards = np.array([(p[0], p[1]) for p in pos], dtype=np.uint64)
    players = np.array([p[2] for p in pos], dtype=np.int8)
    eng = Engine(max_games=slots, max_sims=sims, precision=precision)
    try:
        if evaluator == "net":
            eng.load_state_dict(rp.seeded_state_dict(0))
        kw = dict(evaluator=evaluator, mode=mode, lockstep=True, seed=5)
        if mode == "az":
            kw.update(dirichlet_alpha=0.3, dirichlet_eps=0.25)
        out = []
        for one_stream in (True, False, False):
            traj, stats = eng.selfplay_round(n, eng.params(sims, 1.4, one_stream=one_stream, two_streams=not one_stream, **kw),
                                             first_game_id=9,
                                             start_boards=boards, start_players=players)
            out.append((trajectories_to_numpy(traj).copy(), stats))
    finally:
        eng.close()
    assert out[0][0].tobytes() == out[1][0].tobytes() == out[2][0].tobytes()
    for key in ("games", "plies", "sims", "net_leaves", "terminal_hits", "depth_sum"):
        assert out[0][1][key] == out[1][1][key] == out[2][1][key], key
    assert (out[0][0]["result"] != 2).all()
    if evaluator != "net" and mode == "compat":        # and they are the oracle's games
        for i in (0, 1, 549, 550, 551, 1099, 1100, 1499):
            b1, b2, player = pos[i]
            moves, visits, result, final, _ = co.selfplay_game(sims, 1.4, evaluator, b1, b2, player)
            r = out[1][0][i]
            assert r["plies"]["action"][:r["length"]].tolist() == moves and r["plies"]["visits"][:len(moves)].tolist() == visits

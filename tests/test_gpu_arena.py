"""Two-network arena on the engine (the batched form of nnbattle/tournament/run_tournament.py:12-42):
colour 1 is searched by one weight set, colour 2 by another, all games in lock step.

The oracle side plays the same games one move at a time with the CPU search, handing each
mover its own evaluator.  Fixed evaluators of weight set k hash the position with
me ^ k*0xD6E8FEB86659FD93, so the two-batch plumbing is checked bit for bit without networks.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SET_MIX = 0xD6E8FEB86659FD93
M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def oracle():
    from oracle import c_oracle as co
    from oracle import ref_port as rp
    return co, rp


@pytest.fixture(scope="module")
def arena_engine():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from alphazero_amd.engine import Engine
    eng = Engine(max_games=256, max_sims=200, n_weight_sets=3)
    yield eng
    eng.close()


def _oracle_arena_game(co, evaluators, sims, c_puct, starter):
    """evaluators[colour-1] = callable(me, opp) -> value used by that colour's searches."""
    b1 = b2 = 0
    player = starter
    moves, visits = [], []
    while not co.lib().orc_is_terminal(b1, b2):
        a, v, _ = co.compat_search(b1, b2, player, sims, c_puct, evaluators[player - 1])
        if a is None:
            break
        moves.append(a)
        visits.append(v)
        _, b1, b2 = co.drop(b1, b2, player, a)
        player = 3 - player
    res = co.lib().orc_result(b1, b2) if co.lib().orc_is_terminal(b1, b2) else 2
    return moves, visits, res, (b1, b2)


def _hash_eval(rp, kind, wset):
    fn = rp.hash_value_q8 if kind == "hash_q8" else rp.hash_value_f32
    mix = (wset * SET_MIX) & M64
    return lambda me, opp: fn(me ^ mix, opp)


@pytest.mark.parametrize("kind,sims,lockstep,sets", [("hash_q8", 60, False, (0, 1)), ("hash_q8", 60, True, (0, 1)),
                                                     ("hash_f32", 40, True, (2, 1))])
def test_arena_fixed_evaluators_vs_oracle(arena_engine, oracle, kind, sims, lockstep, sets):
    from alphazero_amd.engine import trajectories_to_numpy
    co, rp = oracle
    eng = arena_engine
    n = 48
    starters = (1 + np.arange(n) % 2).astype(np.int8)
    params = eng.params(sims, 1.4, evaluator=kind, lockstep=lockstep, weight_set=sets[0], weight_set_colour2=sets[1])
    traj, stats = eng.selfplay_round(n, params, start_boards=np.zeros((n, 2), dtype=np.uint64), start_players=starters)
    rec = trajectories_to_numpy(traj)
    evs = [_hash_eval(rp, kind, sets[0]), _hash_eval(rp, kind, sets[1])]
    expect = {s: _oracle_arena_game(co, evs, sims, 1.4, s) for s in (1, 2)}
    # the two evaluators must actually steer differently, or the test would not see a mix-up
    same = _oracle_arena_game(co, [evs[0], evs[0]], sims, 1.4, 1)
    assert same[0] != expect[1][0]
    for i in range(n):
        moves, visits, res, final = expect[int(starters[i])]
        r = rec[i]
        assert r["length"] == len(moves)
        assert r["plies"]["action"][:len(moves)].tolist() == moves
        assert r["plies"]["visits"][:len(moves)].tolist() == visits
        assert r["plies"]["player"][0] == starters[i]
        assert r["result"] == res
        assert (int(r["final_boards"][0]), int(r["final_boards"][1])) == final
    assert stats["games"] == n


def test_arena_same_set_on_both_sides_is_plain_selfplay(arena_engine):
    from alphazero_amd.engine import trajectories_to_numpy
    eng = arena_engine
    n = 32
    starters = (1 + np.arange(n) % 2).astype(np.int8)
    zeros = np.zeros((n, 2), dtype=np.uint64)
    a, _ = eng.selfplay_round(n, eng.params(50, 1.4, evaluator="hash_q8", lockstep=True, weight_set=1, weight_set_colour2=1),
                              start_boards=zeros, start_players=starters)
    a = trajectories_to_numpy(a).copy()
    b, _ = eng.selfplay_round(n, eng.params(50, 1.4, evaluator="hash_q8", lockstep=True, weight_set=1),
                              start_boards=zeros, start_players=starters)
    b = trajectories_to_numpy(b)
    assert a.tobytes() == b.tobytes()


def test_arena_rejects_unknown_or_unloaded_sets(arena_engine):
    from alphazero_amd._lib import NnbError
    eng = arena_engine
    with pytest.raises(NnbError):
        eng.selfplay_round(4, eng.params(10, 1.4, evaluator="hash_q8", weight_set=0, weight_set_colour2=3))
    with pytest.raises(NnbError):
        eng.selfplay_round(4, eng.params(10, 1.4, evaluator="net", weight_set=0, weight_set_colour2=1))


@pytest.mark.parametrize("precision", ["f16", "f32", "f32s"])
def test_arena_two_networks(oracle, precision):
    """Both networks batched on the device.  (a) The same weights loaded as sets 0 and 1 must
    replay single-set self-play byte for byte although the leaves now travel in two batches;
    (b) seeds 0 and 1 against each other follow the CPU game, move for move on the float32
    and float32-faithful paths (the latter: two small batches, whose convolutions run four blocks per board on two
    streams with a scratch area each) and at least through the opening with half inputs (a near tie may flip there)."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    co, rp = oracle
    sd0, sd1 = rp.seeded_state_dict(0), rp.seeded_state_dict(1)
    eng = Engine(max_games=64, max_sims=64, n_weight_sets=2, precision=precision)
    try:
        n, sims = 24, 40
        starters = (1 + np.arange(n) % 2).astype(np.int8)
        zeros = np.zeros((n, 2), dtype=np.uint64)
        eng.load_state_dict(sd0, 0)
        eng.load_state_dict(sd0, 1)
        one, _ = eng.selfplay_round(n, eng.params(sims, 1.4), start_boards=zeros, start_players=starters)
        one = trajectories_to_numpy(one).copy()
        two, _ = eng.selfplay_round(n, eng.params(sims, 1.4, weight_set=0, weight_set_colour2=1),
                                    start_boards=zeros, start_players=starters)
        assert trajectories_to_numpy(two).tobytes() == one.tobytes()

        eng.load_state_dict(sd1, 1)
        traj, _ = eng.selfplay_round(n, eng.params(sims, 1.4, weight_set=0, weight_set_colour2=1),
                                     start_boards=zeros, start_players=starters)
        rec = trajectories_to_numpy(traj)
        nets = [rp.make_net_evaluator(sd0), rp.make_net_evaluator(sd1)]

        def bits_eval(net):
            def f(me, opp):
                from alphazero_amd.game import bits_to_board
                return net(bits_to_board(me, opp), 1)
            return f
        evs = [bits_eval(nets[0]), bits_eval(nets[1])]
        for s in (1, 2):
            moves, visits, res, _final = _oracle_arena_game(co, evs, sims, 1.4, s)
            for i in range(s - 1, n, 2):
                r = rec[i]
                got = r["plies"]["action"][:r["length"]].tolist()
                if precision in ("f32", "f32s"):
                    assert got == moves and r["result"] == res
                    assert r["plies"]["visits"][:len(moves)].tolist() == visits
                else:
                    assert got[:4] == moves[:4]
            print(precision, "starter", s, "cpu game", moves, "device game",
                  rec[s - 1]["plies"]["action"][:rec[s - 1]["length"]].tolist())
    finally:
        eng.close()


@pytest.mark.parametrize("lockstep", [False, True])
def test_arena_az_mode_fixed_evaluators_vs_oracle(arena_engine, oracle, lockstep):
    """The same two-evaluator arena under the PUCT rule (priors and values both come from the
    root mover's evaluator)."""
    import ctypes as C
    from alphazero_amd.engine import trajectories_to_numpy
    co, rp = oracle
    eng = arena_engine
    sims, n = 48, 32

    def az_eval(wset):
        mix = (wset * SET_MIX) & M64
        def f(me, opp, legal):
            pri = (C.c_float * 7)()
            co.lib().orc_fixed_priors(co.EVAL_HASH_Q8, me ^ mix, opp, legal, pri)
            return list(pri), rp.hash_value_q8(me ^ mix, opp)
        return f
    evs = [az_eval(0), az_eval(2)]
    starters = (1 + np.arange(n) % 2).astype(np.int8)
    prm = eng.params(sims, 1.25, evaluator="hash_q8", mode="az", lockstep=lockstep, weight_set=0, weight_set_colour2=2)
    traj, _ = eng.selfplay_round(n, prm, start_boards=np.zeros((n, 2), dtype=np.uint64), start_players=starters)
    rec = trajectories_to_numpy(traj)
    for s in (1, 2):
        b1 = b2 = 0
        player = s
        moves, visits = [], []
        while not co.lib().orc_is_terminal(b1, b2):
            a, v, _ = co.az_search(b1, b2, player, sims, 1.25, evs[player - 1])
            moves.append(a); visits.append(v)
            _, b1, b2 = co.drop(b1, b2, player, a)
            player = 3 - player
        for r in rec[s - 1::2]:
            assert r["plies"]["action"][:r["length"]].tolist() == moves
            assert r["plies"]["visits"][:len(moves)].tolist() == visits
            assert r["result"] == co.lib().orc_result(b1, b2)

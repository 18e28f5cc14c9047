"""NNB_FLAG_EVAL_CACHE: a game's leaf evaluations are filed per position and reused.  The
evaluators are functions of the position, so the games must not change by a byte; only the
number of leaves sent to the evaluator drops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rp():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from oracle import ref_port
    return ref_port


@pytest.fixture(scope="module")
def cached_engine(rp):
    from alphazero_amd.engine import Engine
    eng = Engine(max_games=64, max_sims=80, n_weight_sets=2, eval_cache_slots=1024)
    eng.load_state_dict(rp.seeded_state_dict(0), 0)
    eng.load_state_dict(rp.seeded_state_dict(1), 1)
    yield eng
    eng.close()


@pytest.mark.parametrize("mode,evaluator,sets,sync", [
    ("compat", "hash_q8", {}, False), ("compat", "hash_q8", {}, True), ("az", "hash_f32", {}, False),
    ("compat", "net", {}, False), ("az", "net", {}, False), ("az", "net", {}, True),
    ("compat", "net", {"weight_set": 0, "weight_set_colour2": 1}, False),
    ("az", "hash_q8", {"weight_set": 0, "weight_set_colour2": 1}, False)])
def test_games_are_the_same_with_and_without_the_cache(cached_engine, mode, evaluator, sets, sync):
    from alphazero_amd.engine import trajectories_to_numpy
    eng = cached_engine
    n = 150                                               # more games than slots: generations are exercised
    starters = (1 + np.arange(n) % 2).astype(np.int8)
    zeros = np.zeros((n, 2), dtype=np.uint64)
    extra = dict(dirichlet_alpha=0.4, dirichlet_eps=0.25, seed=21) if mode == "az" else {}
    out = []
    for cache in (False, True):
        prm = eng.params(80, 1.4, evaluator=evaluator, mode=mode, lockstep=True, sync_moves=sync, eval_cache=cache,
                         **extra, **sets)
        traj, stats = eng.selfplay_round(n, prm, start_boards=zeros, start_players=starters)
        out.append((trajectories_to_numpy(traj).copy(), stats))
    (a, sa), (b, sb) = out
    assert a.tobytes() == b.tobytes()
    assert sa["cache_hits"] == 0 and sb["cache_hits"] > 0
    for st in (sa, sb):
        assert st["net_leaves"] + st["terminal_hits"] + st["cache_hits"] == st["sims"]
    assert sb["sims"] == sa["sims"] and sb["net_leaves"] < sa["net_leaves"]
    print(mode, evaluator, "leaves", sa["net_leaves"], "->", sb["net_leaves"], "steps", sa["steps"], "->", sb["steps"])


def test_single_searches_with_the_cache(cached_engine):
    eng = cached_engine
    boards = np.zeros((40, 2), dtype=np.uint64)
    boards[:, 0] = [1 << (7 * (i % 7)) for i in range(40)]
    players = np.full(40, 2, dtype=np.int8)
    for mode in ("compat", "az"):
        a = eng.search(boards, players, eng.params(80, 1.4, evaluator="net", mode=mode))
        b = eng.search(boards, players, eng.params(80, 1.4, evaluator="net", mode=mode, eval_cache=True))
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2].tobytes() == b[2].tobytes()
        assert b[3]["cache_hits"] > 0


def test_cache_flag_needs_a_cache(rp):
    from alphazero_amd._lib import NnbError
    from alphazero_amd.engine import Engine
    eng = Engine(max_games=4, max_sims=16)
    try:
        with pytest.raises(NnbError):
            eng.selfplay_round(4, eng.params(16, 1.4, evaluator="hash_q8", lockstep=True, eval_cache=True))
    finally:
        eng.close()
    with pytest.raises(NnbError):
        Engine(max_games=4, max_sims=16, eval_cache_slots=1000)


def test_agent_level_switch(rp):
    """agent.eval_cache = True: same self-play samples, fewer network rows."""
    from alphazero_amd.agent import AlphaZeroAgent
    from alphazero_amd.selfplay import play_games
    import torch
    torch.manual_seed(0)
    agent = AlphaZeroAgent(7, num_simulations=60, load_model=False)
    plain, st0 = play_games(agent, 3)
    agent.eval_cache = True
    cached, st1 = play_games(agent, 3)
    assert plain.tobytes() == cached.tobytes()
    assert st1["cache_hits"] > 0 and st1["net_leaves"] < st0["net_leaves"]
    from alphazero_amd.game import ConnectFourGame
    agent.eval_cache = False
    a = agent.select_move(ConnectFourGame())
    agent.eval_cache = True
    b = agent.select_move(ConnectFourGame())
    assert a[0] == b[0] and torch.equal(a[1], b[1])

"""Self-play as an endless stream of games (nnb_selfplay_stream): games stay seated between calls, every
slot stays occupied, and what a game plays depends on its id and start position only - so each record
must equal, byte for byte, the record nnb_selfplay_round writes for the same id and start position
(that round is held to the oracle and to the reference's captured games in the other GPU tests)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch


def _stream_records(eng, params, calls, per_call, **kw):
    from alphazero_amd.engine import trajectories_to_numpy
    recs, totals = [], {"games": 0, "sims": 0, "plies": 0}
    for i in range(calls):
        r, st = eng.selfplay_stream(per_call, params if i == 0 else None, restart=(i == 0), **kw)
        assert st["games"] >= per_call and r.shape[0] == st["games"]
        recs.append(trajectories_to_numpy(r).copy())
        for k in totals:
            totals[k] += st[k]
        assert st["finished_total"] == totals["games"]
    return np.concatenate(recs), totals


@pytest.mark.parametrize("evaluator,mode", [("hash_q8", "compat"), ("hash_f32", "az")])
def test_stream_records_equal_round_records(gpu, evaluator, mode):
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.selfplay import random_openings
    slots, sims, n_starts = 48, 40, 100
    eng = Engine(max_games=slots, max_sims=sims)
    try:
        extra = dict(dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=5) if mode == "az" else {}
        params = eng.params(sims, 1.4, evaluator=evaluator, mode=mode, lockstep=True, **extra)
        sb, sp = random_openings(eng, n_starts, 5, seed=3)
        rec, totals = _stream_records(eng, params, calls=4, per_call=70, first_game_id=1000, start_boards=sb, start_players=sp)
        ids = rec["game_id"] - 1000
        assert len(set(ids.tolist())) == len(ids) and ids.min() >= 0            # every game listed once
        assert (rec["result"] != 2).all()
        n = int(ids.max()) + 1
        assert n > n_starts                                                       # start positions wrapped around
        # the same games as one plain round
        sel = np.arange(n) % n_starts
        traj, _ = eng.selfplay_round(n, params, first_game_id=1000, start_boards=sb[sel], start_players=sp[sel])
        ref = trajectories_to_numpy(traj)
        assert rec.tobytes() == ref[ids].tobytes()
        assert totals["plies"] >= int(rec["length"].sum())                        # games in flight have played plies too
        # the plain round ended the stream
        with pytest.raises(Exception, match="restart"):
            eng.selfplay_stream(1)
    finally:
        eng.close()


def test_stream_with_the_network_two_chains(gpu):
    """Network evaluator on enough slots for the two-stream split; the ring wraps (ring_len = 4 x slots)."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.network import seeded_state_dict
    from alphazero_amd.selfplay import random_openings
    slots, sims = 64, 24
    eng = Engine(max_games=slots, max_sims=sims)
    try:
        eng.load_state_dict(seeded_state_dict(0))
        params = eng.params(sims, 1.4, evaluator="net", mode="compat", two_streams=True)
        sb, sp = random_openings(eng, 512, 6, seed=11)
        rec, totals = _stream_records(eng, params, calls=6, per_call=60, start_boards=sb, start_players=sp,
                                      ring_len=4 * slots)
        assert totals["games"] > 4 * slots                                        # the ring was reused
        ids = rec["game_id"]
        assert len(set(ids.tolist())) == len(ids)
        n = int(ids.max()) + 1
        sel = np.arange(n) % 512
        traj, st = eng.selfplay_round(n, params, first_game_id=0, start_boards=sb[sel], start_players=sp[sel])
        ref = trajectories_to_numpy(traj)
        assert rec.tobytes() == ref[ids].tobytes()
    finally:
        eng.close()


def test_stream_argument_checks(gpu):
    from alphazero_amd.engine import Engine
    eng = Engine(max_games=16, max_sims=8)
    try:
        p = eng.params(8, 1.4, evaluator="hash_q8", lockstep=True)
        with pytest.raises(Exception, match="restart"):
            eng.selfplay_stream(1)
        with pytest.raises(Exception, match="ring"):
            eng.selfplay_stream(1, p, restart=True, ring_len=17)
        with pytest.raises(Exception, match="LOCKSTEP"):
            eng.selfplay_stream(1, eng.params(8, 1.4, evaluator="hash_q8"), restart=True)
        r, st = eng.selfplay_stream(0, p, restart=True)
        assert st["games"] == 0 and r.shape[0] == 0
        r, st = eng.selfplay_stream(5)
        assert st["games"] >= 5
        with pytest.raises(Exception, match="fit the ring"):
            eng.selfplay_stream(8 * 16 - 2 * 16 + 1)
        with pytest.raises(Exception, match="differ"):
            eng.selfplay_stream(1, eng.params(8, 2.0, evaluator="hash_q8", lockstep=True))
    finally:
        eng.close()


def test_stream_of_finished_start_positions_ends(gpu):
    """Start positions that are all finished: every game is recorded as it is seated, no slot ever holds a game,
    and the call must come back instead of seating for ever - with an error, because the slots recorded far more
    games than the ring holds before they gave up."""
    from alphazero_amd.engine import Engine
    eng = Engine(max_games=16, max_sims=8)
    try:
        p = eng.params(8, 1.4, evaluator="hash_q8", lockstep=True)
        won = np.array([[0b1111, 0b1110000000]], dtype=np.uint64)          # colour 1 has four in the first column
        with pytest.raises(Exception, match="overwritten"):                 # recorded, never played; every slot gave up
            eng.selfplay_stream(96, p, restart=True, start_boards=won, start_players=np.array([2], dtype=np.int8))
        with pytest.raises(Exception, match="restart"):                     # nothing is seated any more
            eng.selfplay_stream(1)
    finally:
        eng.close()


def test_weights_reloaded_between_calls_of_a_stream(gpu):
    """The training loop's pattern: self-play, new weights, more self-play, without emptying the slots.  Games seated
    after the reload must equal the games a plain round plays with the new weights; games finished before it those of
    the old weights; the ones in flight across the reload are only required to be legal games."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.network import seeded_state_dict
    from alphazero_amd.selfplay import random_openings
    slots, sims, n_starts = 32, 16, 256
    eng = Engine(max_games=slots, max_sims=sims)
    try:
        sd_a, sd_b = seeded_state_dict(0), seeded_state_dict(1)
        eng.load_state_dict(sd_a)
        params = eng.params(sims, 1.4, evaluator="net", mode="compat")
        sb, sp = random_openings(eng, n_starts, 6, seed=4)
        r1, st1 = eng.selfplay_stream(40, params, restart=True, start_boards=sb, start_players=sp)
        seated_at_reload = st1["finished_total"] + slots            # ids below this were seated under the old weights
        eng.load_state_dict(sd_b)
        r2, st2 = eng.selfplay_stream(120)
        old, new = trajectories_to_numpy(r1).copy(), trajectories_to_numpy(r2).copy()
        assert (old["result"] != 2).all() and (new["result"] != 2).all()
        assert (old["game_id"] < seated_at_reload).all()
        late = new[new["game_id"] >= seated_at_reload]
        assert len(late) > 20
        n = int(max(old["game_id"].max(), new["game_id"].max())) + 1
        sel = np.arange(n) % n_starts
        ref_b = trajectories_to_numpy(eng.selfplay_round(n, params, start_boards=sb[sel], start_players=sp[sel])[0]).copy()
        assert late.tobytes() == ref_b[late["game_id"]].tobytes()
        eng.load_state_dict(sd_a)
        ref_a = trajectories_to_numpy(eng.selfplay_round(n, params, start_boards=sb[sel], start_players=sp[sel])[0]).copy()
        assert old.tobytes() == ref_a[old["game_id"]].tobytes()
        assert (ref_a[late["game_id"]]["plies"].tobytes() != late["plies"].tobytes())      # the weights matter
    finally:
        eng.close()


def test_stream_with_the_evaluation_cache_plays_the_same_games(gpu):
    """NNB_FLAG_EVAL_CACHE on a stream: a slot's table is emptied when the slot takes its next game, so records
    still equal the uncached plain round's byte for byte while fewer leaves reach the network."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.network import seeded_state_dict
    from alphazero_amd.selfplay import random_openings
    slots, sims = 40, 32
    eng = Engine(max_games=slots, max_sims=sims, eval_cache_slots=256)
    try:
        eng.load_state_dict(seeded_state_dict(0))
        sb, sp = random_openings(eng, 300, 5, seed=8)
        cached = eng.params(sims, 1.4, evaluator="net", mode="compat", eval_cache=True)
        rec, totals = _stream_records(eng, cached, calls=3, per_call=60, start_boards=sb, start_players=sp)
        assert totals["games"] > slots
        n = int(rec["game_id"].max()) + 1
        sel = np.arange(n) % 300
        traj, st = eng.selfplay_round(n, eng.params(sims, 1.4, evaluator="net", mode="compat"), start_boards=sb[sel], start_players=sp[sel])
        ref = trajectories_to_numpy(traj)
        assert rec.tobytes() == ref[rec["game_id"]].tobytes()
        assert st["cache_hits"] == 0
    finally:
        eng.close()

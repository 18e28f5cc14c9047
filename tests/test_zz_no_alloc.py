"""nnb_search_static - the entry point that allocates nothing - against nnb_search, through the C ABI.

Two calls in a row on the same GPU, the second with fewer roots and fewer simulations: the static
search state keeps the first call's contents, so anything the second call fails to reset would show."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _positions(n, seed):
    from oracle import c_oracle as co
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        b1 = b2 = 0
        player = 1
        ok = True
        for _ in range(int(rng.randint(0, 30))):
            m = co.legal_mask(b1, b2)
            cols = [c for c in range(7) if (m >> c) & 1]
            _, b1, b2 = co.drop(b1, b2, player, int(rng.choice(cols)))
            player = 3 - player
            if co.lib().orc_is_terminal(b1, b2):
                ok = False
                break
        if ok:
            out.append((b1, b2, player))
    return out


def _static_search(lib, torch, params, boards, players, sides=None, noise=None):
    dev = torch.device("cuda", torch.cuda.current_device())
    n = len(boards)
    b = torch.from_numpy(np.asarray(boards, dtype=np.uint64).view(np.int64)).to(dev)
    pl = torch.from_numpy(np.asarray(players, dtype=np.int8)).to(dev)
    sd = None if sides is None else torch.from_numpy(np.asarray(sides, dtype=np.int8)).to(dev)
    nz = None if noise is None else torch.from_numpy(np.asarray(noise, dtype=np.float32)).to(dev)
    # outputs pre-filled so that a row the call does not write is noticed
    visits = torch.full((max(n, 1), 7), -7, dtype=torch.int32, device=dev)
    action = torch.full((max(n, 1),), -7, dtype=torch.int8, device=dev)
    rootv = torch.full((max(n, 1),), -7.0, dtype=torch.float64, device=dev)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    before = torch.cuda.memory_allocated(dev)
    rc = lib.nnb_search_static(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.byref(params), n, ptr(b),
                               ptr(pl), ptr(sd), ptr(nz), ptr(visits), ptr(action), ptr(rootv))
    assert torch.cuda.memory_allocated(dev) == before
    torch.cuda.synchronize(dev)
    return rc, visits.cpu().numpy()[:n], action.cpu().numpy()[:n], rootv.cpu().numpy()[:n]


def test_static_search_equals_engine_search_twice_in_a_row():
    import torch
    from alphazero_amd import _lib
    from alphazero_amd.engine import Engine
    assert torch.cuda.is_available()
    lib = _lib.load()
    eng = Engine(max_games=256, max_sims=800, precision="f16")
    try:
        pos = _positions(256, seed=2718)
        boards = np.array([(p[0], p[1]) for p in pos], dtype=np.uint64)
        players = np.array([p[2] for p in pos], dtype=np.int8)
        sides = np.array([1 + (i % 2) for i in range(len(pos))], dtype=np.int8)
        rng = np.random.RandomState(5)
        noise = rng.dirichlet([0.3] * 7, size=len(pos)).astype(np.float32)
        calls = [
            # (roots, params, sides, noise): the first call is the largest the entry point accepts
            (256, eng.params(800, 1.4, evaluator="hash_f32"), sides, None),
            (37, eng.params(60, 1.4, evaluator="hash_q8"), None, None),
            (200, eng.params(150, 1.5, evaluator="hash_q8", mode="az", dirichlet_eps=0.25), None, noise),
            (5, eng.params(90, 1.1, evaluator="rollout", mode="az", dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=99,
                           rng_game_base=40, rng_move_index=3), None, None),
            (64, eng.params(200, 1.4, evaluator="rollout", seed=7), None, None),
            (1, eng.params(1, 1.4, evaluator="const"), None, None),
        ]
        for n, prm, sd, nz in calls:
            rc, v, a, rv = _static_search(lib, torch, prm, boards[:n], players[:n], None if sd is None else sd[:n],
                                          None if nz is None else nz[:n])
            assert rc == 0, lib.nnb_last_error()
            v2, a2, rv2, _ = eng.search(boards[:n], players[:n], prm, root_noise=None if nz is None else nz[:n],
                                        side_players=None if sd is None else sd[:n])
            assert (v == v2).all() and (a == a2).all() and (rv == rv2).all(), (n, prm.sims)
            assert (v.sum(axis=1) == prm.sims - 1).all()
    finally:
        eng.close()


def test_static_search_refuses_what_does_not_fit():
    import torch
    from alphazero_amd import _lib
    from alphazero_amd.engine import Engine
    lib = _lib.load()
    eng = Engine(max_games=4, max_sims=8, precision="f16")
    try:
        boards = np.zeros((300, 2), dtype=np.uint64)
        players = np.ones(300, dtype=np.int8)
        ok = eng.params(10, 1.4, evaluator="const")
        for n, prm, word in ((257, ok, b"NNB_STATIC_MAX_ROOTS"), (4, eng.params(801, 1.4, evaluator="const"), b"NNB_STATIC_MAX_SIMS"),
                             (4, eng.params(10, 1.4, evaluator="net"), b"fixed evaluators"),
                             (4, eng.params(10, 1.4, evaluator="const", lockstep=True), b"no flags")):
            rc, v, a, _ = _static_search(lib, torch, prm, boards[:n], players[:n])
            assert rc == -1 and word in lib.nnb_last_error()
            assert (v == -7).all() and (a == -7).all()           # nothing ran
        rc, v, a, _ = _static_search(lib, torch, ok, boards[:0], players[:0])
        assert rc == 0                                           # an empty call is a no-op
    finally:
        eng.close()


def test_small_static_search_does_not_shrink_the_engines_lds_budget():
    """The fused tree kernel's dynamic-LDS ceiling is a property of the kernel on the device: an engine raises it
    for an 800-simulation search, a small static search must not lower it again for the engine's next one."""
    import torch
    from alphazero_amd import _lib
    from alphazero_amd.engine import Engine
    lib = _lib.load()
    eng = Engine(max_games=8, max_sims=800, precision="f16")
    try:
        pos = _positions(8, seed=31)
        boards = np.array([(p[0], p[1]) for p in pos], dtype=np.uint64)
        players = np.array([p[2] for p in pos], dtype=np.int8)
        big = eng.params(800, 1.4, evaluator="hash_q8", mode="az")
        first = eng.search(boards, players, big)
        rc, v, a, _ = _static_search(lib, torch, eng.params(20, 1.4, evaluator="hash_q8"), boards, players)
        assert rc == 0, lib.nnb_last_error()
        again = eng.search(boards, players, big)
        assert (first[0] == again[0]).all() and (first[0].sum(axis=1) == 799).all()
    finally:
        eng.close()


def test_static_search_from_two_streams():
    """A call on a second stream right after a call on the first: the one-time set-up (log table) was waited for
    by the first call, so the second stream finds it filled whatever it is ordered against."""
    import torch
    from alphazero_amd import _lib
    from alphazero_amd.engine import Engine
    lib = _lib.load()
    eng = Engine(max_games=64, max_sims=100, precision="f16")
    try:
        pos = _positions(64, seed=77)
        boards = np.array([(p[0], p[1]) for p in pos], dtype=np.uint64)
        players = np.array([p[2] for p in pos], dtype=np.int8)
        prm = eng.params(100, 1.4, evaluator="hash_f32")
        want = eng.search(boards, players, prm)
        for st in (torch.cuda.Stream(), torch.cuda.Stream()):
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                rc, v, a, rv = _static_search(lib, torch, prm, boards, players)
            assert rc == 0, lib.nnb_last_error()
            assert (v == want[0]).all() and (a == want[1]).all() and (rv == want[2]).all()
    finally:
        eng.close()

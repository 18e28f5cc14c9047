"""The tree step finishes the forward pass of its own pending leaf (heads of network.py:40-41,48-49
and the mask / renormalisation of agent_code.py:199-208) instead of reading values and priors a
separate heads kernel wrote.  Both forms share their arithmetic, so whole self-play rounds must
come out byte for byte the same."""
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


@pytest.mark.parametrize("precision", ["f32s", "f16", "f32"])
@pytest.mark.parametrize("mode", ["compat", "az"])
def test_fused_and_separate_heads_agree(rp, precision, mode):
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    eng = Engine(max_games=48, max_sims=32, n_weight_sets=2, precision=precision)
    try:
        eng.load_state_dict(rp.seeded_state_dict(0), 0)
        eng.load_state_dict(rp.seeded_state_dict(1), 1)
        n = 80                                            # more games than slots: refill is exercised too
        starters = (1 + np.arange(n) % 2).astype(np.int8)
        zeros = np.zeros((n, 2), dtype=np.uint64)
        extra = dict(dirichlet_alpha=0.3, dirichlet_eps=0.25, seed=5) if mode == "az" else {}
        for sets in ({}, {"weight_set": 0, "weight_set_colour2": 1}):
            out = []
            for separate in (False, True):
                prm = eng.params(32, 1.4, evaluator="net", mode=mode, separate_heads=separate, **extra, **sets)
                traj, stats = eng.selfplay_round(n, prm, start_boards=zeros, start_players=starters)
                out.append((trajectories_to_numpy(traj).copy(), stats))
            assert out[0][0].tobytes() == out[1][0].tobytes()
            assert out[0][1] == out[1][1]
            if mode == "az":
                assert len({r["plies"]["action"][:r["length"]].tobytes() for r in out[0][0]}) > 20
        # single searches too (the root value is the sum of the leaf values the heads produced)
        boards = np.zeros((48, 2), dtype=np.uint64)
        players = np.ones(48, dtype=np.int8)
        a = eng.search(boards, players, eng.params(32, 1.4, evaluator="net", mode=mode, **extra))
        b = eng.search(boards, players, eng.params(32, 1.4, evaluator="net", mode=mode, separate_heads=True, **extra))
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2].tobytes() == b[2].tobytes()
    finally:
        eng.close()


def test_search_backs_up_the_value_the_network_entry_point_returns(rp):
    """Float32-faithful path: a one-simulation search in the reference's rule backs up exactly the network's value of
    the root position as seen by the mover (agent_code.py:195-212).  The search (leaf batch through the lock-step
    forward pass, value head only) and nnb_net_forward (all heads, hidden activations copied out) must give the same
    float for every one of 100 positions, 33 whole row tiles of the linear layer's small form and one partial."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.selfplay import random_openings
    eng = Engine(max_games=100, max_sims=32, precision="f32s")
    try:
        eng.load_state_dict(rp.seeded_state_dict(0), 0)
        sb, sp = random_openings(eng, 100, 6, seed=21)
        one = eng.search(sb, sp, eng.params(1, 1.4, evaluator="net", mode="compat"))
        side = np.where(sp[:, None] == 1, sb, sb[:, ::-1]).astype(np.uint64)
        _, value, _ = eng.net_forward(side)
        assert np.asarray(one[2], dtype=np.float64).tobytes() == value.cpu().numpy().astype(np.float64).tobytes()
        # the retired two-launch flag is accepted and changes nothing
        a = eng.search(sb, sp, eng.params(32, 1.4, evaluator="net"))
        b = eng.search(sb, sp, eng.params(32, 1.4, evaluator="net", separate_convs=True))
        assert (a[0] == b[0]).all() and a[2].tobytes() == b[2].tobytes()
    finally:
        eng.close()


@pytest.mark.parametrize("precision", ["f32s", "f16"])
def test_small_batch_and_large_batch_kernels_agree_bit_for_bit(rp, precision):
    """Small batches run latency-oriented forms of the network kernels: at most 128 positions with half inputs (one
    board per convolution block), at most 128 float32-faithful positions (four convolution blocks per board, meeting
    through float32 activations in memory) and at most 256 (one wave per linear-layer block); larger ones the
    throughput forms.  A position's outputs and hidden activations must not depend on which form evaluated it."""
    import torch
    from alphazero_amd.engine import Engine
    from alphazero_amd.selfplay import random_openings
    eng = Engine(max_games=400, max_sims=8, precision=precision)
    try:
        eng.load_state_dict(rp.seeded_state_dict(0), 0)
        sb, sp = random_openings(eng, 400, 6, seed=9)
        side = np.where(sp[:, None] == 1, sb, sb[:, ::-1]).astype(np.uint64)
        legal = eng.game_ops(sb)[1]
        big = eng.net_forward(side, legal)                          # 400 rows: the large-batch kernels
        acts_big = [eng.activations(layer, 100).clone() for layer in (2, 3, 4)]
        for n in (1, 2, 3, 31, 64, 65, 100, 128, 129, 250):
            small = eng.net_forward(side[:n], legal[:n])            # the small-batch kernels (f16: up to 128 rows)
            for a, b in zip(big, small):
                assert torch.equal(a[:n], b), n
            for layer, ref in zip((2, 3, 4), acts_big):
                assert torch.equal(eng.activations(layer, min(n, 100)), ref[:min(n, 100)]), (n, layer)
    finally:
        eng.close()


@pytest.mark.parametrize("rows", [1500, 2500, 3500])
def test_every_row_tile_count_of_the_linear_layer_agrees_with_the_small_form(rp, rows):
    """The large form of the float32-faithful linear layer gives a block 32, 64, 96 or 128 rows depending on the batch
    (1500, 2500 and 3500 rows take 64, 96 and 128; 400 rows, in the test above, 32), each a different instantiation of
    its loop.  The same 200 positions, repeated to fill the batch, must come out as the small form evaluates them."""
    import torch
    from alphazero_amd.engine import Engine
    from alphazero_amd.selfplay import random_openings
    eng = Engine(max_games=rows, max_sims=2, precision="f32s")
    try:
        eng.load_state_dict(rp.seeded_state_dict(0), 0)
        sb, sp = random_openings(eng, 200, 6, seed=21)
        side200 = np.where(sp[:, None] == 1, sb, sb[:, ::-1]).astype(np.uint64)
        small_logp, small_value, _ = eng.net_forward(side200)       # 200 rows: one wave per linear-layer block
        logp, value, _ = eng.net_forward(side200[np.arange(rows) % 200])
        idx = torch.arange(rows, device=logp.device) % 200
        assert torch.equal(logp, small_logp[idx]) and torch.equal(value, small_value[idx])
    finally:
        eng.close()


def test_rows_beyond_four_gigabytes_of_activations(rp):
    """One activation row of the float32-faithful linear layer is 46 080 bytes, so from row 93 207 on a row's byte
    offset no longer fits 32 bits.  The kernels address a row block through a 64-bit base: positions evaluated in the
    last rows of a 93 500-row batch must come out exactly as in a small batch."""
    import torch
    from alphazero_amd.engine import Engine
    from alphazero_amd.selfplay import random_openings
    n = 93500
    eng = Engine(max_games=n, max_sims=2, precision="f32s")
    try:
        eng.load_state_dict(rp.seeded_state_dict(0), 0)
        sb, sp = random_openings(eng, 300, 6, seed=13)
        side300 = np.where(sp[:, None] == 1, sb, sb[:, ::-1]).astype(np.uint64)
        side = side300[np.arange(n) % 300]
        logp, value, _ = eng.net_forward(side)
        small_logp, small_value, _ = eng.net_forward(side300)
        idx = torch.arange(n, device=logp.device) % 300
        assert torch.equal(logp, small_logp[idx]) and torch.equal(value, small_value[idx])
        assert (n - 1) * 46080 >= 2 ** 32
    finally:
        eng.close()


def test_linear_layer_hand_off_under_uneven_load(rp):
    """The two K halves of a linear-layer tile meet inside the launch: the block that arrives second waits for the
    first one's partial sums.  With another stream keeping part of the chip busy the two arrive far apart, on
    whatever CUs are free; outputs and hidden activations must still be those of a quiet run, bit for bit, for
    batch sizes that use both kernel forms and all row-tile counts."""
    import torch
    from alphazero_amd.engine import Engine
    from alphazero_amd.selfplay import random_openings
    eng = Engine(max_games=3000, max_sims=4, precision="f32s")
    try:
        eng.load_state_dict(rp.seeded_state_dict(0), 0)
        sb, sp = random_openings(eng, 3000, 6, seed=3)
        side = np.where(sp[:, None] == 1, sb, sb[:, ::-1]).astype(np.uint64)
        side_t = torch.from_numpy(side.view(np.int64)).to(eng.device)
        sizes = (3000, 2048, 1100, 700, 257, 256, 33)
        quiet = {}
        for n in sizes:
            lp, v, _ = eng.net_forward(side_t[:n])
            quiet[n] = (lp.clone(), v.clone(), eng.activations(4, min(n, 64)).clone())
        noise_stream = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device=eng.device)
        for rep in range(6):
            with torch.cuda.stream(noise_stream):
                for _ in range(6 + 5 * rep):
                    a = torch.tanh(a @ a * 1e-3)                   # a few milliseconds of somebody else's kernels
            for n in sizes:
                lp, v, _ = eng.net_forward(side_t[:n])
                assert torch.equal(lp, quiet[n][0]) and torch.equal(v, quiet[n][1]), (rep, n)
                assert torch.equal(eng.activations(4, min(n, 64)), quiet[n][2]), (rep, n)
        torch.cuda.synchronize()
        # a search reads the engine's count of waits that ran out and fails loudly if there was one
        eng.search(sb[:8], sp[:8], eng.params(4, 1.4, evaluator="net"))
    finally:
        eng.close()

"""The drop-in Python API (AlphaZeroAgent / self_play / run_tournament) on the GPU engine,
against the golden vectors captured from the reference and the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    return torch


def test_select_move_replays_golden_games(torch_mod, search_golden):
    """The loop of train_alpha_zero.py:37-48 through the drop-in classes reproduces the reference's
    moves and root visit counts ply by ply under the fixed evaluators."""
    from nnbattle.agents.alphazero.utils import initialize_agent
    from nnbattle.game import ConnectFourGame
    for g in search_golden["games"]:
        if g["sims"] > 200:
            continue
        agent = initialize_agent(action_dim=7, state_dim=2, use_gpu=False, num_simulations=g["sims"],
                                 c_puct=g["c_puct"], load_model=False)
        agent.evaluator = g["evaluator"]
        game = ConnectFourGame()
        game.reset()
        agent.current_player = 1
        moves, visits = [], []
        while not game.is_terminal() and len(moves) < len(g["moves"]):
            action, probs = agent.select_move(game)
            assert probs.dtype == torch_mod.float32 and tuple(probs.shape) == (7,)
            total = g["sims"] - 1
            visits.append([int(round(float(p) * total)) for p in probs])
            moves.append(action)
            game.make_move(action)
            agent.current_player = game.current_player
        assert moves == g["moves"] and visits == g["visits"]
        if g["complete"]:
            assert game.get_result() == g["result"]


def test_select_move_reads_the_agents_own_side(torch_mod, sided_golden):
    """select_move with agent.current_player different from game.current_player: the planes follow the
    agent's attribute (agent_code.py:85-87), the stones placed during descent the game's mover
    (agent_code.py:167,176) - vectors captured from the reference with the two set apart."""
    from nnbattle.agents.alphazero.utils import initialize_agent
    from nnbattle.game import ConnectFourGame
    from oracle import ref_port as rp
    cases = [c for c in sided_golden["positions"] if c["evaluator"] == "hash_f32" and c["sims"] == 60]
    agent = initialize_agent(action_dim=7, state_dim=2, use_gpu=False, num_simulations=60, c_puct=1.4,
                             load_model=False)
    agent.evaluator = "hash_f32"
    assert any(c["side"] != c["player"] for c in cases)
    for c in cases:
        game = ConnectFourGame()
        game.board = rp.board_from_str(c["board"])
        game.current_player = c["player"]
        agent.current_player = c["side"]
        action, probs = agent.select_move(game)
        assert action == c["action"]
        assert [int(round(float(p) * 59)) for p in probs] == c["visits"]
        assert agent.current_player == c["side"] and game.current_player == c["player"]


def test_select_move_on_finished_game_returns_none(torch_mod):
    from alphazero_amd.agent import initialize_agent
    from alphazero_amd.game import ConnectFourGame
    agent = initialize_agent(load_model=False, num_simulations=20)
    agent.evaluator = "const"
    game = ConnectFourGame()
    for _ in range(4):
        game.make_move(0)
        game.current_player = 1
    assert game.is_terminal()
    assert agent.select_move(game) is None                       # agent_code.py:116-118
    one = initialize_agent(load_model=False, num_simulations=1)
    one.evaluator = "const"
    with pytest.raises(ZeroDivisionError):                       # 0/0 visits, agent_code.py:229
        one.select_move(ConnectFourGame())


def test_batched_self_play_matches_reference_samples(torch_mod, search_golden):
    from alphazero_amd.agent import initialize_agent
    from alphazero_amd.selfplay import self_play
    from oracle import ref_port as rp
    g = next(x for x in search_golden["games"] if x["evaluator"] == "hash_q8" and x["sims"] == 100)
    agent = initialize_agent(load_model=False, num_simulations=100)
    agent.evaluator = "hash_q8"
    memory = self_play(agent, 5)
    assert len(memory) == 5 and len(agent.memory) == 5
    final = rp.board_from_str(g["final_board"])
    mover_after = 1 if len(g["moves"]) % 2 == 0 else 2
    for planes, probs, result in memory:
        assert result == g["result"]
        assert (planes.cpu().numpy() == rp.planes(final, mover_after)).all()
        assert float(probs.sum()) == 0.0 and tuple(probs.shape) == (7,)


def test_net_agent_weight_refresh_and_real_net_game(torch_mod, net_golden_meta):
    torch = torch_mod
    from alphazero_amd.agent import initialize_agent
    from alphazero_amd.game import ConnectFourGame
    from oracle import ref_port as rp
    torch.manual_seed(0)
    agent = initialize_agent(load_model=False, num_simulations=100)      # seed-0 random init, as captured
    g = net_golden_meta["game_s100"]
    game = ConnectFourGame()
    moves = []
    for _ in range(4):
        agent.current_player = game.current_player
        a, probs = agent.select_move(game)
        moves.append(a)
        game.make_move(a)
    assert moves == g["moves"][:4]
    # new weights in agent.model must reach the engine without any explicit call
    agent.model.load_state_dict(rp.seeded_state_dict(1))
    game = ConnectFourGame()
    agent.current_player = 1
    a1, p1 = agent.select_move(game)
    ev = rp.make_net_evaluator(rp.seeded_state_dict(1))
    a_ref, v_ref = rp.compat_search(rp.new_board(), 1, 100, 1.4, ev)
    assert a1 == a_ref and [int(round(float(p) * 99)) for p in p1] == v_ref


def test_tournament_between_two_engine_agents(torch_mod, tmp_path):
    from alphazero_amd.agent import initialize_agent
    from alphazero_amd.tournament import run_tournament
    a = initialize_agent(load_model=False, num_simulations=40)
    b = initialize_agent(load_model=False, num_simulations=40)
    a.evaluator, b.evaluator = "hash_q8", "hash_f32"
    res = run_tournament([a, b], num_games=4, results_dir=str(tmp_path))
    assert set(res) == {"AlphaZeroAgent_0", "AlphaZeroAgent_1", "draws"} and sum(res.values()) == 4


@pytest.mark.parametrize("order", [0, 1])
def test_batched_minimax_vs_alphazero_tournament_equals_the_move_by_move_protocol(torch_mod, tmp_path, order):
    """The reference's own pairing (run_tournament.py:44-48).  All games advanced together on the
    device must end exactly as when each agent is asked for one move at a time on the host."""
    from alphazero_amd.agent import AlphaZeroAgent
    from alphazero_amd.game import ConnectFourGame
    from alphazero_amd.minimax import MinimaxAgent
    from alphazero_amd.tournament import _names, mixed_arena, play_match, run_tournament
    az = AlphaZeroAgent(7, num_simulations=40, load_model=False)
    az.evaluator = "hash_q8"
    agents = [MinimaxAgent(depth=3), az] if order == 0 else [az, MinimaxAgent(depth=3)]
    names = _names(agents)
    game = ConnectFourGame()
    want = [play_match(agents, names, game, i % 2) for i in range(6)]
    got = mixed_arena(agents, 6)
    assert ["draws" if w is None else names[w] for w in got] == want
    tally = run_tournament(agents, num_games=6, results_dir=str(tmp_path))
    assert tally == {n: want.count(n) for n in names} | {"draws": want.count("draws")}


def test_per_ply_training_samples_follow_the_cpu_game(torch_mod):
    """self_play_samples: (planes from the mover's side, visit distribution, z) per ply, the format
    agent_code.py:254-267 intends, against the CPU search's game."""
    from alphazero_amd.agent import AlphaZeroAgent
    from alphazero_amd.selfplay import self_play_samples
    from oracle import c_oracle as co
    from oracle import ref_port as rp
    agent = AlphaZeroAgent(7, num_simulations=60, load_model=False)
    agent.evaluator = "hash_q8"
    samples = self_play_samples(agent, 2)
    moves, visits, result, _final, _ = co.selfplay_game(60, 1.4, "hash_q8")
    assert len(samples) == 2 * len(moves)
    board = rp.new_board()
    player = 1
    for k, (mv, vis) in enumerate(zip(moves, visits)):
        planes, probs, z = samples[k]
        assert (planes.numpy() == rp.planes(board, player)).all()
        want = np.array(vis, dtype=np.float64) / sum(vis)
        assert np.allclose(probs.numpy(), want.astype(np.float32))
        assert z == (0 if result in (0, None) else (1 if (result == 1) == (player == 1) else -1))
        _, player = rp.make_move(board, player, mv)
    assert all((a[0] == b[0]).all() and a[2] == b[2] for a, b in zip(samples[:len(moves)], samples[len(moves):]))


def test_weights_file_round_trip_into_the_engine(torch_mod, tmp_path, monkeypatch):
    """save_agent_model -> load_agent_model (utils.py:33-61: torch.save of the state dict, torch.load with
    weights_only=True) -> engine: an agent that loaded the file searches exactly as the agent that wrote it,
    and its first move is the CPU reference search's."""
    import alphazero_amd.agent as agent_mod
    from nnbattle.agents.alphazero.utils import initialize_agent, load_agent_model, save_agent_model
    from nnbattle.game import ConnectFourGame
    from oracle import ref_port as rp
    torch = torch_mod
    torch.manual_seed(123)
    writer = initialize_agent(action_dim=7, state_dim=2, use_gpu=False, num_simulations=60, c_puct=1.4, load_model=False)
    path = tmp_path / "alphazero_model_final.pth"
    save_agent_model(writer, str(path))
    monkeypatch.setattr(agent_mod, "MODEL_PATH", str(path))
    torch.manual_seed(999)                                       # the reader's own initial weights differ
    reader = initialize_agent(action_dim=7, state_dim=2, use_gpu=False, num_simulations=60, c_puct=1.4, load_model=True)
    assert reader.model_loaded
    for k, v in writer.model.state_dict().items():
        assert torch.equal(v, reader.model.state_dict()[k]), k
    game = ConnectFourGame()
    game.reset()
    game.make_move(3); game.make_move(2); game.make_move(3)
    writer.current_player = reader.current_player = game.current_player
    aw, pw = writer.select_move(game)
    ar, pr = reader.select_move(game)
    assert aw == ar and torch.equal(pw, pr)
    # against the CPU restatement of the reference search with the same file's weights
    sd = torch.load(str(path), map_location="cpu", weights_only=True)
    a_cpu, v_cpu = rp.compat_search(np.array(game.board), game.current_player, 60, 1.4, rp.make_net_evaluator(sd))
    assert ar == a_cpu
    assert [int(round(float(p) * 59)) for p in pr] == list(v_cpu)
    # a missing file is the reference's error (utils.py:50-51)
    monkeypatch.setattr(agent_mod, "MODEL_PATH", str(tmp_path / "nothing.pth"))
    with pytest.raises(FileNotFoundError):
        load_agent_model(reader)


def test_per_ply_samples_built_on_the_device_equal_the_host_loop(torch_mod):
    """nnb_trajectory_samples against the host loop of self_play_samples on the same records: planes, visit
    distributions and value targets, bit for bit, over games of every outcome (wins of both colours, draws are
    rare: an unfinished record is forged to cover z = 0) and an empty record."""
    from alphazero_amd.engine import Engine, trajectories_to_numpy
    from alphazero_amd.selfplay import _planes, random_openings
    torch = torch_mod
    eng = Engine(max_games=64, max_sims=24)
    try:
        sb, sp = random_openings(eng, 200, 6, seed=2)
        traj, _ = eng.selfplay_round(200, eng.params(24, 1.4, evaluator="hash_f32"), start_boards=sb, start_players=sp)
        rec = trajectories_to_numpy(traj).copy()
        assert {1, -1} <= set(rec["result"].tolist())
        rec["result"][5] = 2                                        # a cut-off game: z = 0 for all its plies
        rec["length"][7] = 0                                        # a game that started finished: no samples
        dev = torch.from_numpy(rec.view(np.uint8).reshape(200, -1)).cuda()
        planes, probs, z = eng.trajectory_samples(dev)
        P = int(rec["length"].sum())
        assert planes.shape == (P, 2, 6, 7) and probs.shape == (P, 7) and z.shape == (P,)
        planes, probs, z = planes.cpu().numpy(), probs.cpu().numpy(), z.cpu().numpy()
        s = 0
        for r in rec:
            res = int(r["result"])
            for i in range(int(r["length"])):
                ply = r["plies"][i]
                me = int(ply["player"])
                vis = ply["visits"].astype(np.float64)
                want = (vis / vis.sum()).astype(np.float32)
                assert planes[s].tobytes() == _planes(int(ply["colour_boards"][0]), int(ply["colour_boards"][1]), me).tobytes()
                assert probs[s].tobytes() == want.tobytes()
                assert z[s] == (0 if res in (0, 2) else (1 if (res == 1) == (me == 1) else -1))
                s += 1
        assert s == P
        e = eng.trajectory_samples(dev[:0])
        assert e[0].shape[0] == 0 and e[2].shape[0] == 0
    finally:
        eng.close()

"""Host-side mirror of the reference API (alphazero_amd.game / mcts / agent / selfplay /
tournament and the nnbattle import paths), checked on CPU against the golden vectors."""
import json
import math
import os

import numpy as np
import pytest
import torch

from alphazero_amd.game import (AI_PIECE, COLUMN_COUNT, EMPTY, PLAYER_PIECE, ROW_COUNT, ConnectFourGame,
                                bits_to_board, board_to_bits)
from alphazero_amd.mcts import MCTSNode
from oracle import ref_port as rp


def _board(s):
    return rp.board_from_str(s)


def _check_game(game, want):
    assert game.get_valid_locations() == want["legal"]
    assert game.is_terminal() == want["terminal"] == game.is_terminal_node()
    assert game.get_reward() == want["reward"]
    assert game.get_result() == want["result"]
    assert game.get_winner() == want["winner"]
    assert game.check_win(1) == want["win1"] and game.check_win(2) == want["win2"]
    assert game.is_board_full() == want["full"]


def test_game_matches_golden_special_boards(rules_golden):
    for sp in rules_golden["special"]:
        g = ConnectFourGame()
        g.board = _board(sp["board"])
        _check_game(g, sp)
        b1, b2 = board_to_bits(g.board)
        assert (b1, b2) == rp.bitboards(g.board)
        assert (bits_to_board(b1, b2) == g.board).all()
        for player in (1, 2):
            assert g.get_win_positions(player) == rp.win_positions(g.board, player)


def test_game_matches_golden_playouts(rules_golden):
    for game in rules_golden["games"]:
        g = ConnectFourGame()
        ply = 0
        _check_game(g, game["plies"][0])
        for mv in game["moves"]:
            before = g.current_player
            assert g.make_move(mv["col"]) == mv["ok"]
            if not mv["ok"]:
                assert g.current_player == before
                continue
            ply += 1
            want = game["plies"][ply]
            assert rp.board_to_str(g.board) == want["board"] and g.current_player == want["player"]
            _check_game(g, want)


class TestReferenceGameTests:
    """The nine tests of nnbattle/game/tests/test_connect_four_game.py, restated."""

    def test_initial_state(self):
        g = ConnectFourGame()
        assert (g.board == EMPTY).all() and g.current_player == PLAYER_PIECE
        assert g.board.dtype == np.int8 and g.board.shape == (ROW_COUNT, COLUMN_COUNT)

    def test_make_move_valid(self):
        g = ConnectFourGame()
        assert g.make_move(3) is True
        assert g.board[5][3] == PLAYER_PIECE and g.current_player == AI_PIECE

    def test_make_move_invalid(self):
        g = ConnectFourGame()
        for _ in range(6):
            assert g.make_move(0)
        assert g.make_move(0) is False
        assert g.current_player == PLAYER_PIECE

    def test_is_terminal_win(self):
        g = ConnectFourGame()
        for _ in range(4):
            assert g.make_move(0)
            g.current_player = PLAYER_PIECE
        assert g.is_terminal() and g.get_winner() == PLAYER_PIECE

    def test_is_terminal_draw(self, rules_golden):
        sp = next(s for s in rules_golden["special"] if s["name"] == "reference_test_draw_board")
        g = ConnectFourGame()
        g.board = _board(sp["board"])
        assert g.is_terminal() and g.get_winner() == EMPTY
        assert not g.check_win(AI_PIECE) and not g.check_win(PLAYER_PIECE)

    def test_score_position_center(self):
        g = ConnectFourGame()
        g.make_move(COLUMN_COUNT // 2)
        g.make_move(COLUMN_COUNT // 2)
        assert g.score_position(AI_PIECE) > 0

    def test_score_position_win(self):
        g = ConnectFourGame()
        for col in range(4):
            g.make_move(col)
            g.make_move(col)
        assert g.score_position(AI_PIECE) > 0

    def test_board_type_after_move(self):
        g = ConnectFourGame()
        g.make_move(3)
        assert isinstance(g.board, np.ndarray)

    def test_is_valid_location_assertion(self):
        g = ConnectFourGame()
        with pytest.raises(AssertionError):
            g.board = "not a numpy array"
            g.is_valid_location(3)


def test_step_and_copies():
    g = ConnectFourGame()
    board, reward, done, info = g.step(2)
    assert board[5][2] == 1 and reward == 0.0 and done is False and info == {}
    board[5][2] = 7
    assert g.board[5][2] == 1                       # outputs are copies
    for _ in range(5):
        g.step(2)
    board, reward, done, info = g.step(2)            # seventh stone in the column
    assert reward == -10.0 and done is True and info == {"invalid_move": True}
    twin = g.copy()
    twin.make_move(0)
    assert g.board[5][0] == 0
    assert (g.reset() == 0).all() and g.current_player == 1
    assert g.get_next_open_row(0) == 5


def _object_search(board, player, sims, c_puct, evaluator):
    """The reference's loop (agent_code.py:159-229) driven through the host classes."""
    root = MCTSNode(parent=None, action=None, env=None)
    root.visits = 1
    for _ in range(sims):
        node = root
        g = ConnectFourGame()
        g.board = board.copy()
        g.current_player = player
        while not node.is_leaf() and not g.is_terminal():
            node = node.best_child(c_puct=c_puct)
            g.make_move(node.action)
            g.current_player = 2 if g.current_player == 1 else 1
        if g.is_terminal():
            node.backpropagate(g.get_reward())
            continue
        legal = g.get_valid_locations()
        node.expand(None, legal)
        node.backpropagate(evaluator(g.board, player))
    visits = [0] * 7
    for ch in root.children.values():
        visits[ch.action] = ch.visits
    return visits


def test_host_classes_reproduce_golden_searches(search_golden):
    cases = [c for c in search_golden["positions"] if c["sims"] == 50][:30]
    for case in cases:
        v = _object_search(_board(case["board"]), case["player"], case["sims"], case["c_puct"],
                           rp.EVALUATORS[case["evaluator"]])
        assert v == case["visits"]


def test_mcts_node_interface():
    root = MCTSNode()
    assert root.is_leaf() and root.visits == 0 and root.value == 0.0 and root.parent is None
    root.visits = 1
    root.expand([0.5, 0.5], [2, 4])
    assert list(root.children) == [2, 4] and not root.is_leaf()
    assert root.best_child(1.4).action == 2                 # tie -> first (lowest) column
    root.children[4].backpropagate(0.5)
    assert root.visits == 2 and root.value == 0.5 and root.children[4].visits == 1
    want = 0.5 / 2 + 1.4 * math.sqrt(math.log(3) / 2)
    assert root.score(root.children[4], 1.4) == want


def test_agent_surface_and_preprocess():
    import inspect
    from alphazero_amd.agent import (MODEL_PATH, Agent, AlphaZeroAgent, initialize_agent, load_agent_model)
    sig = inspect.signature(initialize_agent)
    assert list(sig.parameters) == ["action_dim", "state_dim", "use_gpu", "num_simulations", "c_puct", "load_model"]
    assert [p.default for p in sig.parameters.values()] == [7, 2, False, 800, 1.4, True]
    assert list(inspect.signature(AlphaZeroAgent.__init__).parameters)[1:] == list(sig.parameters)
    agent = initialize_agent(load_model=False, num_simulations=50)
    assert isinstance(agent, Agent)
    for attr in ("state_dim", "action_dim", "device", "model_loaded", "load_model_flag", "num_simulations",
                 "c_puct", "current_player", "memory", "model", "logger"):
        assert hasattr(agent, attr)
    for meth in ("preprocess", "load_model", "save_model", "select_move", "act", "mcts_simulate", "self_play",
                 "perform_training", "log_gpu_stats"):
        assert callable(getattr(agent, meth))
    assert agent.memory == [] and agent.current_player == 1 and agent.c_puct == 1.4
    g = ConnectFourGame()
    g.make_move(3); g.make_move(4)
    for me in (1, 2):
        agent.current_player = me
        x = agent.preprocess(g.board)
        assert x.dtype == torch.float32 and tuple(x.shape) == (2, 6, 7)
        assert (x.numpy() == rp.planes(g.board, me)).all()
    assert sorted(agent.model.state_dict()) == sorted(rp.seeded_state_dict(0))
    assert MODEL_PATH.endswith("alphazero_model_final.pth")
    with pytest.raises(FileNotFoundError):
        load_agent_model(agent)
    with pytest.raises(FileNotFoundError):
        initialize_agent(load_model=True)


def test_model_container_matches_reference_forward():
    from alphazero_amd.network import Connect4Net, seeded_state_dict
    sd = seeded_state_dict(0)
    net = Connect4Net(2, 7)
    net.load_state_dict(rp.seeded_state_dict(0))
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "net_seed0.npz"))
    x = torch.from_numpy(np.stack([rp.planes(b, int(m)) for b, m in zip(z["boards"][:32], z["me"][:32])]))
    with torch.no_grad():
        lp, v = net(x)
    assert tuple(lp.shape) == (32, 7) and tuple(v.shape) == (32, 1)
    np.testing.assert_allclose(lp.numpy(), z["log_policy"][:32], atol=2e-5, rtol=0)
    np.testing.assert_allclose(v.numpy()[:, 0], z["value"][:32], atol=2e-5, rtol=0)
    assert all(torch.equal(sd[k], net.state_dict()[k]) for k in sd)
    with pytest.raises(AssertionError):
        net(torch.zeros(1, 3, 6, 7))


def test_nnbattle_import_paths():
    import nnbattle
    from nnbattle.agents import AlphaZeroAgent, MinimaxAgent
    from nnbattle.agents.alphazero import initialize_agent, load_agent_model, save_agent_model
    from nnbattle.agents.alphazero.agent_code import AlphaZeroAgent as A2
    from nnbattle.agents.alphazero.mcts import MCTSNode as M2
    from nnbattle.agents.alphazero.network import Connect4Net
    from nnbattle.agents.alphazero.train.train_alpha_zero import self_play
    from nnbattle.agents.alphazero.utils import MODEL_PATH, deepcopy_env
    from nnbattle.agents.base_agent import Agent
    from nnbattle.game import ConnectFourGame as G2
    from nnbattle.game.connect_four_game import PLAYER_PIECE as P
    from nnbattle.tournament import run_tournament
    assert A2 is AlphaZeroAgent and M2 is MCTSNode and G2 is ConnectFourGame and P == 1
    assert set(nnbattle.__all__) == {"AlphaZeroAgent", "MinimaxAgent", "ConnectFourGame", "run_tournament"}
    g = ConnectFourGame()
    g.make_move(1)
    twin = deepcopy_env(g)
    twin.make_move(1)
    assert g.board[4][1] == 0 and twin.board[4][1] == 2


class _ColumnAgent:
    def __init__(self, order):
        self.order = order

    def select_move(self, game):
        for c in self.order:
            if game.is_valid_location(c):
                return c
        return None


class _LeftAgent(_ColumnAgent):
    pass


def test_tournament_protocol(tmp_path):
    from alphazero_amd.tournament import run_tournament
    a, b = _LeftAgent([0, 1, 2, 3, 4, 5, 6]), _ColumnAgent([6, 5, 4, 3, 2, 1, 0])
    res = run_tournament([a, b], num_games=4, results_dir=str(tmp_path))
    # whoever starts stacks four in its own column first
    assert res == {"_LeftAgent": 2, "_ColumnAgent": 2, "draws": 0}
    assert json.load(open(tmp_path / "tournament_results.json")) == res
    same = run_tournament([_LeftAgent([0]), _LeftAgent([1])], num_games=2, results_dir=str(tmp_path))
    assert set(same) == {"_LeftAgent_0", "_LeftAgent_1", "draws"}
    stuck = run_tournament([_ColumnAgent([]), b], num_games=2, results_dir=str(tmp_path))
    assert stuck["draws"] >= 1                           # no move offered counts as a draw


def test_minimax_takes_the_win_and_blocks():
    from alphazero_amd.minimax import MinimaxAgent
    agent = MinimaxAgent(depth=2)
    g = ConnectFourGame()
    for _ in range(3):
        g.make_move(2); g.make_move(5)                 # colour 1 has three in column 2, colour 2 three in 5
    assert agent.select_move(g) == 2                   # colour 1 to move wins at once
    g.make_move(0)
    assert agent.select_move(g) == 5                   # colour 2 to move wins at once
    assert 0 <= MinimaxAgent(depth=4).select_move(ConnectFourGame()) <= 6


def test_shard_range_partitions_games():
    from alphazero_amd.selfplay import shard_range
    for n, w in ((32768, 8), (10, 3), (5, 8), (0, 4)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_minimax_mirror_agrees_with_the_oracle_search():
    """alphazero_amd.minimax.MinimaxAgent (host Python) against oracle/c4_oracle.c::orc_minimax, which
    the reference's captured vectors pin (tests/golden/minimax.json)."""
    import numpy as np
    from alphazero_amd.game import ConnectFourGame, bits_to_board
    from alphazero_amd.minimax import MinimaxAgent
    from oracle import c_oracle as co
    rng = np.random.RandomState(12)
    checked = 0
    while checked < 40:
        b1 = b2 = 0
        player = 1
        for _ in range(int(rng.randint(0, 30))):
            if co.lib().orc_is_terminal(b1, b2):
                break
            m = co.legal_mask(b1, b2)
            _, b1, b2 = co.drop(b1, b2, player, int(rng.choice([c for c in range(7) if (m >> c) & 1])))
            player = 3 - player
        if co.lib().orc_is_terminal(b1, b2):
            continue
        g = ConnectFourGame()
        g.board = bits_to_board(b1, b2)
        g.current_player = player
        for depth in (1, 2, 3):
            want, _ = co.minimax(b1, b2, player, depth)
            assert MinimaxAgent(depth=depth).select_move(g) == want
        checked += 1

"""Capture golden vectors from the imported reference into tests/golden/.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the build
container); the fixtures it writes are data (inputs + expected outputs) and are
committed, so no test ever needs the reference at run time.

Import recipe (reference files are neither copied nor modified):
  * a stub `pytorch_lightning` is registered before `import nnbattle`, because
    lightning_module.py:3, data_module.py:3 and train_alpha_zero.py:8-9 import it
    at package import time and it is not installed; none of it is on the path
    being captured;
  * `agent_code.deepcopy` is bound to copy.deepcopy afterwards, because
    agent_code.py:166 calls deepcopy without importing it.

Usage:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden.py
"""
from __future__ import annotations

import copy
import hashlib
import json
import logging
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import ref_port as rp  # noqa: E402  (hash definitions + encodings only)


def import_reference():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    pl.LightningDataModule = object
    pl.Trainer = object
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.ModelCheckpoint = object
    pl.callbacks = cb
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.callbacks"] = cb
    import nnbattle  # noqa: F401
    import nnbattle.agents.alphazero.agent_code as agent_code
    agent_code.deepcopy = copy.deepcopy
    logging.disable(logging.CRITICAL)
    from nnbattle.game.connect_four_game import ConnectFourGame
    from nnbattle.agents.alphazero.utils import initialize_agent
    return ConnectFourGame, initialize_agent


ConnectFourGame, initialize_agent = import_reference()


# ---------------------------------------------------------------- fake models
class FixedModel(torch.nn.Module):
    """Stands in for agent.model: uniform log-policy + a value that is a pure
    function of the input planes (so the search sees a fixed evaluator)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind

    def forward(self, x):
        assert x.shape == (1, 2, 6, 7)
        lp = torch.full((1, 7), float(np.log(1.0 / 7.0)))
        if self.kind == "const":
            return lp, torch.zeros(1, 1)
        p = x[0].numpy()
        me = opp = 0
        for r in range(6):
            for c in range(7):
                bit = 1 << (c * 7 + (5 - r))
                if p[0, r, c] == 1.0:
                    me |= bit
                if p[1, r, c] == 1.0:
                    opp |= bit
        if self.kind == "hash_q8":
            v = rp.hash_value_q8(me, opp)
        elif self.kind == "hash_f32":
            v = rp.hash_value_f32(me, opp)
        else:
            raise ValueError(self.kind)
        return lp, torch.tensor([[v]], dtype=torch.float32)


def make_agent(kind_or_seed, sims):
    if isinstance(kind_or_seed, int):
        torch.manual_seed(kind_or_seed)
        agent = initialize_agent(action_dim=7, state_dim=2, use_gpu=False,
                                 num_simulations=sims, c_puct=1.4, load_model=False)
    else:
        agent = initialize_agent(action_dim=7, state_dim=2, use_gpu=False,
                                 num_simulations=sims, c_puct=1.4, load_model=False)
        agent.model = FixedModel(kind_or_seed)
    return agent


def play_game(agent, max_plies=None):
    """The loop body of train_alpha_zero.py:37-48, recording per-ply visits."""
    game = ConnectFourGame()
    game.reset()
    agent.current_player = 1
    moves, visits = [], []
    while not game.is_terminal():
        a, probs = agent.select_move(game)
        total = agent.num_simulations - 1
        v = [int(round(float(p) * total)) for p in probs]
        assert sum(v) == total, (v, total)
        moves.append(int(a))
        visits.append(v)
        game.make_move(a)
        agent.current_player = game.current_player
        if max_plies is not None and len(moves) >= max_plies:
            break
    return moves, visits, game.get_result(), game.board.copy()


def search_position(agent, board, player, side=None):
    """One select_move from (board, player to move).  `side` is what agent.current_player holds
    during the search (agent_code.py:85-87 reads it for the planes, independently of the mover
    that agent_code.py:167 copies from the game); None = the mover, as self_play keeps it."""
    game = ConnectFourGame()
    game.board = board.copy()
    game.current_player = player
    agent.current_player = player if side is None else side
    a, probs = agent.select_move(game)
    total = agent.num_simulations - 1
    v = [int(round(float(p) * total)) for p in probs]
    assert sum(v) == total
    return int(a), v


# ---------------------------------------------------------------- rule vectors
def snapshot(game):
    return {
        "board": rp.board_to_str(game.board),
        "player": int(game.current_player),
        "legal": [int(c) for c in game.get_valid_locations()],
        "terminal": bool(game.is_terminal()),
        "reward": float(game.get_reward()),
        "result": game.get_result(),
        "winner": int(game.get_winner()),
        "win1": bool(game.check_win(1)),
        "win2": bool(game.check_win(2)),
        "full": bool(game.is_board_full()),
    }


def capture_rules():
    rng = np.random.RandomState(20250131)
    games = []
    for g in range(64):
        game = ConnectFourGame()
        plies = [snapshot(game)]
        moves = []
        # some games continue past a win to exercise double-win / full boards
        stop_at_terminal = g % 4 != 3
        while True:
            legal = game.get_valid_locations()
            if not legal or (stop_at_terminal and game.is_terminal()):
                break
            # occasionally try a full column: must fail without flipping (tests :43-51)
            col = int(rng.choice(legal))
            full_cols = [c for c in range(7) if c not in legal]
            if full_cols and rng.rand() < 0.3:
                bad = int(rng.choice(full_cols))
                before = (game.board.copy(), game.current_player)
                ok = game.make_move(bad)
                assert ok is False and game.current_player == before[1]
                assert (game.board == before[0]).all()
                moves.append({"col": bad, "ok": False})
            ok = game.make_move(col)
            assert ok is True
            moves.append({"col": col, "ok": True})
            plies.append(snapshot(game))
        games.append({"moves": moves, "plies": plies})

    special = []
    draw = np.array([            # literal of test_connect_four_game.py:74-81
        [1, 1, 1, 2, 2, 2, 1],
        [2, 2, 2, 1, 1, 1, 2],
        [1, 1, 1, 2, 2, 2, 1],
        [2, 2, 2, 1, 1, 1, 2],
        [1, 1, 1, 2, 2, 2, 1],
        [2, 2, 2, 1, 1, 1, 2]], dtype=np.int8)
    g = ConnectFourGame()
    g.board = draw
    special.append({"name": "reference_test_draw_board", **snapshot(g)})
    # every one of the 69 windows, for both colours, on an otherwise empty board
    for player in (1, 2):
        for (r, c, dr, dc) in all_windows():
            b = np.zeros((6, 7), dtype=np.int8)
            for i in range(4):
                b[r + i * dr][c + i * dc] = player
            g = ConnectFourGame()
            g.board = b
            special.append({"name": f"window_p{player}_{r}_{c}_{dr}_{dc}", **snapshot(g)})
    # three-in-a-row and wrap-around non-wins
    for player in (1, 2):
        for (r, c, dr, dc) in all_windows():
            b = np.zeros((6, 7), dtype=np.int8)
            for i in range(3):
                b[r + i * dr][c + i * dc] = player
            b[r + 3 * dr][c + 3 * dc] = 3 - player
            g = ConnectFourGame()
            g.board = b
            special.append({"name": f"blocked_p{player}_{r}_{c}_{dr}_{dc}", **snapshot(g)})
    wrap = np.zeros((6, 7), dtype=np.int8)
    wrap[5, 5] = wrap[5, 6] = wrap[4, 0] = wrap[4, 1] = 1     # row wrap must not count
    g = ConnectFourGame()
    g.board = wrap
    special.append({"name": "row_wrap", **snapshot(g)})
    wrap2 = np.zeros((6, 7), dtype=np.int8)
    wrap2[0, 2] = wrap2[1, 2] = wrap2[5, 3] = wrap2[4, 3] = 2  # column wrap must not count
    g = ConnectFourGame()
    g.board = wrap2
    special.append({"name": "col_wrap", **snapshot(g)})
    return {"games": games, "special": special}


def all_windows():
    out = []
    for r in range(6):
        for c in range(4):
            out.append((r, c, 0, 1))
    for c in range(7):
        for r in range(3):
            out.append((r, c, 1, 0))
    for r in range(3):
        for c in range(4):
            out.append((r, c, 1, 1))
    for r in range(3, 6):
        for c in range(4):
            out.append((r, c, -1, 1))
    assert len(out) == 69
    return out


# ---------------------------------------------------------------- search vectors
def random_positions(n, seed, min_plies=0, max_plies=30):
    """Non-terminal positions reached by random legal play (board, player)."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        game = ConnectFourGame()
        k = int(rng.randint(min_plies, max_plies + 1))
        ok = True
        for _ in range(k):
            legal = game.get_valid_locations()
            game.make_move(int(rng.choice(legal)))
            if game.is_terminal():
                ok = False
                break
        if ok:
            out.append((game.board.copy(), int(game.current_player)))
    return out


def capture_search():
    games, positions = [], []
    t0 = time.time()
    for kind in ("const", "hash_q8", "hash_f32"):
        for sims in (100, 200, 400, 800):
            if sims == 800 and kind != "hash_q8":
                max_plies = 4                     # keep the capture bounded
            elif sims >= 400:
                max_plies = 10 if kind != "hash_q8" else None
            else:
                max_plies = None
            agent = make_agent(kind, sims)
            moves, visits, result, board = play_game(agent, max_plies)
            games.append({"evaluator": kind, "sims": sims, "c_puct": 1.4, "moves": moves,
                          "visits": visits, "result": result, "complete": result is not None,
                          "final_board": rp.board_to_str(board)})
            print(f"[search] {kind} S={sims}: {len(moves)} plies result={result} "
                  f"({time.time() - t0:.0f}s)", flush=True)
    pos = random_positions(24, seed=7, min_plies=1, max_plies=34)
    for kind in ("const", "hash_q8", "hash_f32"):
        for sims in (50, 200):
            agent = make_agent(kind, sims)
            for (b, p) in pos:
                a, v = search_position(agent, b, p)
                positions.append({"evaluator": kind, "sims": sims, "c_puct": 1.4,
                                  "board": rp.board_to_str(b), "player": p,
                                  "action": a, "visits": v})
        print(f"[search] positions {kind} done ({time.time() - t0:.0f}s)", flush=True)
    # other exploration constants, to pin the score formula's use of c_puct
    for c in (0.5, 2.5):
        agent = make_agent("hash_q8", 150)
        agent.c_puct = c
        for (b, p) in pos[:8]:
            a, v = search_position(agent, b, p)
            positions.append({"evaluator": "hash_q8", "sims": 150, "c_puct": c,
                              "board": rp.board_to_str(b), "player": p, "action": a, "visits": v})
    return {"games": games, "positions": positions}


def capture_sided():
    """Searches in which agent.current_player differs from the colour to move: the fixed
    evaluators (bit-exact) and the seed-0 network (float dependent).  Also what the reference's
    own tournament loop does, which never updates agent.current_player (run_tournament.py:21-22)."""
    positions = []
    pos = random_positions(16, seed=23, min_plies=1, max_plies=30)
    for kind in ("hash_q8", "hash_f32"):
        for sims in (60, 200):
            agent = make_agent(kind, sims)
            for (b, p) in pos:
                for side in (1, 2):
                    a, v = search_position(agent, b, p, side)
                    positions.append({"evaluator": kind, "sims": sims, "c_puct": 1.4, "board": rp.board_to_str(b),
                                      "player": p, "side": side, "action": a, "visits": v})
        print(f"[sided] {kind} done", flush=True)
    agent = make_agent(0, 60)
    for (b, p) in pos[:12]:
        side = 3 - p
        a, v = search_position(agent, b, p, side)
        positions.append({"evaluator": "net_seed0", "sims": 60, "c_puct": 1.4, "board": rp.board_to_str(b),
                          "player": p, "side": side, "action": a, "visits": v})
    print("[sided] net done", flush=True)
    return {"positions": positions}


def capture_timing():
    """The CPU restatement (oracle/ref_port.py) timed next to the imported reference, same
    container, same weights, same searches: how far the port's speed is from the reference's."""
    # this container's 8 virtual CPUs are shared with other tenants: with 8 intra-op threads a batch-1
    # forward pass takes 30x longer than with 2-4 (spinning workers lose their core), so both sides run
    # with 4 threads; the ratio, not the absolute rate, is what this capture is for
    torch.set_num_threads(4)
    sd = rp.seeded_state_dict(0)
    ev = rp.make_net_evaluator(sd)
    out = {"cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(), "legs": []}
    for sims, plies in ((100, 8), (200, 6)):
        agent = make_agent(0, sims)
        game = ConnectFourGame()
        game.reset()
        agent.current_player = 1
        board, player = rp.new_board(), 1
        t_ref = t_port = 0.0
        for _ in range(plies):
            t0 = time.perf_counter()
            a, _ = agent.select_move(game)
            t_ref += time.perf_counter() - t0
            t0 = time.perf_counter()
            a2, _ = rp.compat_search(board, player, sims, 1.4, ev)
            t_port += time.perf_counter() - t0
            assert a == a2, (a, a2)
            game.make_move(a)
            agent.current_player = game.current_player
            _, player = rp.make_move(board, player, a)
        out["legs"].append({"sims_per_move": sims, "plies": plies,
                            "reference_sims_per_s": sims * plies / t_ref, "port_sims_per_s": sims * plies / t_port,
                            "port_over_reference": t_ref / t_port})
        print(f"[timing] S={sims}: reference {sims * plies / t_ref:.1f} sims/s, port {sims * plies / t_port:.1f} sims/s",
              flush=True)
    return out


# ---------------------------------------------------------------- net vectors
def state_dict_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def capture_net():
    agent = make_agent(0, 100)
    sd = agent.model.state_dict()
    digest = state_dict_digest(sd)
    assert digest == state_dict_digest(rp.seeded_state_dict(0)), "seeded init mismatch"
    n_params = sum(v.numel() for v in sd.values())
    pos = random_positions(256, seed=11, min_plies=0, max_plies=38)
    boards = np.stack([b for b, _ in pos]).astype(np.int8)
    me = np.array([p for _, p in pos], dtype=np.int8)
    logp = np.zeros((len(pos), 7), dtype=np.float32)
    val = np.zeros((len(pos),), dtype=np.float32)
    priors = np.zeros((len(pos), 7), dtype=np.float64)
    for i, (b, p) in enumerate(pos):
        agent.current_player = p
        x = agent.preprocess(b).unsqueeze(0)                 # agent_code.py:195
        with torch.no_grad():
            lp, v = agent.model(x)                           # agent_code.py:198
        logp[i] = lp[0].numpy()
        val[i] = v.item()
        policy = torch.exp(lp).cpu().numpy().flatten()       # agent_code.py:199
        legal = [c for c in range(7) if b[0][c] == 0]
        f = np.zeros(7)                                      # agent_code.py:202-208
        f[legal] = policy[legal]
        if f.sum() > 0:
            f /= f.sum()
        else:
            f[legal] = 1.0 / len(legal)
        priors[i] = f
    np.savez_compressed(os.path.join(GOLDEN, "net_seed0.npz"), boards=boards, me=me,
                        log_policy=logp, value=val, priors=priors)
    # one self-play game with the real net (float dependent: pinned as agreement, not bits)
    t0 = time.time()
    moves, visits, result, board = play_game(agent)
    dt = time.time() - t0
    meta = {"seed": 0, "sha256": digest, "n_params": int(n_params),
            "game_s100": {"moves": moves, "visits": visits, "result": result,
                          "final_board": rp.board_to_str(board)},
            "reference_timing_here": {"sims_per_move": 100, "plies": len(moves), "seconds": dt,
                                      "games_per_s": 1.0 / dt,
                                      "sims_per_s": 100 * len(moves) / dt,
                                      "cpu_count": os.cpu_count(),
                                      "torch_threads": torch.get_num_threads()}}
    print(f"[net] seed-0 game: {len(moves)} plies in {dt:.1f}s", flush=True)
    return meta


# ---------------------------------------------------------------- minimax opponent
def capture_minimax():
    """score_position (connect_four_game.py:215-278) on random boards, and MinimaxAgent.minimax
    (minimax/agent_code.py:34-89) on positions where colour 2 is to move - the only case in which
    the reference agent searches for the side it moves (it always maximises for AI_PIECE = 2)."""
    import math
    import random
    from nnbattle.agents.minimax import MinimaxAgent
    random.seed(0)
    out = {"scores": [], "searches": []}
    for board, player in random_positions(300, seed=404, min_plies=0, max_plies=38):
        g = ConnectFourGame()
        g.board = board.copy()
        out["scores"].append({"board": rp.board_to_str(board), "score1": int(g.score_position(1)),
                              "score2": int(g.score_position(2))})
    def enc(x):
        return "inf" if x == math.inf else "-inf" if x == -math.inf else int(x)
    pool = [(b, p) for b, p in random_positions(400, seed=505, min_plies=1, max_plies=36) if p == 2]
    for depth, count in ((1, 60), (2, 60), (3, 40), (4, 40)):
        for board, player in pool[:count]:
            g = ConnectFourGame()
            g.board = board.copy()
            g.current_player = 2
            score, col = MinimaxAgent(depth=depth).minimax(g, depth, -math.inf, math.inf, True)
            out["searches"].append({"board": rp.board_to_str(board), "depth": depth, "score": enc(score),
                                    "column": None if col is None else int(col),
                                    # every move loses inside the horizon: the column is random.choice
                                    "column_is_random": score == -math.inf})
        print(f"[minimax] depth {depth} done", flush=True)
    return out


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    which = sys.argv[1:] or ["rules", "search", "net"]
    if "rules" in which:
        with open(os.path.join(GOLDEN, "rules.json"), "w") as f:
            json.dump(capture_rules(), f, separators=(",", ":"))
        print("rules.json written", flush=True)
    if "search" in which:
        with open(os.path.join(GOLDEN, "search.json"), "w") as f:
            json.dump(capture_search(), f, separators=(",", ":"))
        print("search.json written", flush=True)
    if "sided" in which:
        with open(os.path.join(GOLDEN, "search_sided.json"), "w") as f:
            json.dump(capture_sided(), f, separators=(",", ":"))
        print("search_sided.json written", flush=True)
    if "timing" in which:
        with open(os.path.join(GOLDEN, "port_vs_reference_timing.json"), "w") as f:
            json.dump(capture_timing(), f, separators=(",", ":"))
        print("port_vs_reference_timing.json written", flush=True)
    if "minimax" in which:
        with open(os.path.join(GOLDEN, "minimax.json"), "w") as f:
            json.dump(capture_minimax(), f, separators=(",", ":"))
        print("minimax.json written", flush=True)
    if "net" in which:
        with open(os.path.join(GOLDEN, "net_seed0.json"), "w") as f:
            json.dump(capture_net(), f, separators=(",", ":"))
        print("net_seed0.json written", flush=True)


if __name__ == "__main__":
    main()

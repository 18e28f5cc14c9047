"""Agent-versus-agent arena with the interface of nnbattle/tournament/run_tournament.py:12-42.

Same protocol: `num_games` games, the starting agent alternates with the game index, the
first agent of the pair plays colour 1, a win is credited to the winner, anything else
(no legal move, full board) counts as a draw, and the tally is written as JSON to
tournament/results/tournament_results.json.  Agents are asked for a move with the game
object (what AlphaZeroAgent.select_move and MinimaxAgent.select_move both take) and may
return either a column or a (column, distribution) pair.  Tallies are keyed by class name,
with an index appended when two agents share a class.

Two engine-backed AlphaZeroAgents with the same search settings do not take turns on the
host at all: `arena` plays every game of the tournament concurrently on one engine that
holds both networks, each search's leaves batched with the other games' leaves that the
same network judges (`run_tournament` takes that route by itself).  The reference's own pairing,
MinimaxAgent against AlphaZeroAgent (run_tournament.py:44-48), is batched too (`mixed_arena`).
"""
from __future__ import annotations

import json
import os

import numpy as np

from .game import AI_PIECE, PLAYER_PIECE, ConnectFourGame

ARENA_CONCURRENT_GAMES = 4096    # games in flight; a longer tournament refills slots as games end


def _names(agents):
    base = [a.__class__.__name__ for a in agents]
    return [n if base.count(n) == 1 else f"{n}_{i}" for i, n in enumerate(base)]


def _column(move):
    if isinstance(move, (tuple, list)):
        move = move[0]
    return None if move is None else int(move)


def play_match(agents, names, game, starter: int, keep_agent_sides: bool = False):
    """One game; returns the winning agent's name or 'draws'.  An AlphaZeroAgent encodes positions
    from its `current_player` attribute (agent_code.py:85-87): by default that is set to the colour
    the agent plays, as self_play does; keep_agent_sides=True leaves it alone, which is what
    run_tournament.py:21-22 does (the attribute then keeps whatever the caller last gave it)."""
    game.reset()
    seat = starter
    while not game.is_terminal_node():
        agent = agents[seat]
        piece = PLAYER_PIECE if seat == 0 else AI_PIECE
        game.current_player = piece
        if hasattr(agent, "current_player") and not keep_agent_sides:
            agent.current_player = piece
        col = _column(agent.select_move(game))
        if col is None or not game.is_valid_location(col):
            return "draws"
        game.drop_piece(game.get_next_open_row(col), col, piece)
        if game.check_win(piece):
            return names[seat]
        seat = 1 - seat
    return "draws"


def arena_capable(agents) -> bool:
    """True when the pair can be played batched on one engine: two AlphaZeroAgents with the
    same simulation count, exploration constant and evaluator kind."""
    from .agent import AlphaZeroAgent
    if len(agents) != 2 or not all(isinstance(a, AlphaZeroAgent) for a in agents):
        return False
    a, b = agents
    return (a.num_simulations == b.num_simulations and a.c_puct == b.c_puct and a.evaluator == b.evaluator
            and a.precision == b.precision and a.num_simulations >= 2)


def arena(agents, num_games: int, engine=None, first_game_id: int = 0, keep_agent_sides: bool = False):
    """Plays `num_games` games between two AlphaZeroAgents on the GPU, all games at once.

    agents[0] plays colour 1 and agents[1] colour 2; game i is started by agents[i % 2]
    (run_tournament.py:17-19).  Returns (wins of agents[0], wins of agents[1], draws, records)
    with `records` the engine's trajectory records (engine.TRAJECTORY_DTYPE).  A game that
    ends with a full board, or in which the mover has no move, is a draw
    (run_tournament.py:30-35).  Each search encodes its leaves from the searching agent's own colour;
    with keep_agent_sides=True from that agent's `current_player` attribute as it stands
    (run_tournament.py:21-22 never updates it).
    """
    from .engine import TRAJECTORY_DTYPE, Engine, trajectories_to_numpy
    if not arena_capable(agents):
        raise ValueError("arena needs two AlphaZeroAgents with equal num_simulations, c_puct, evaluator and precision")
    a, b = agents
    for ag in agents:
        if ag.load_model_flag and not ag.model_loaded:
            ag.load_model()
    n_round = min(int(num_games), ARENA_CONCURRENT_GAMES)
    own = engine is None
    if own:
        engine = Engine(max_games=max(n_round, 1), max_sims=a.num_simulations, n_weight_sets=2, precision=a.precision)
    try:
        if a.evaluator == "net":
            engine.load_state_dict(a.model.state_dict(), weight_set=0)
            engine.load_state_dict(b.model.state_dict(), weight_set=1)
        sides = (2 if a.current_player == 2 else 1, 2 if b.current_player == 2 else 1) if keep_agent_sides else (0, 0)
        params = engine.params(a.num_simulations, a.c_puct, evaluator=a.evaluator, mode="compat", weight_set=0,
                               weight_set_colour2=1, side_of_colour=sides)
        n = int(num_games)
        records = np.zeros(0, dtype=TRAJECTORY_DTYPE)
        if n > 0:
            starters = (1 + (np.arange(n) % 2)).astype(np.int8)
            traj, _stats = engine.selfplay_round(n, params, first_game_id=first_game_id,
                                                 start_boards=np.zeros((n, 2), dtype=np.uint64), start_players=starters)
            records = trajectories_to_numpy(traj)
    finally:
        if own:
            engine.close()
    res = records["result"]
    wins_a, wins_b = int((res == 1).sum()), int((res == -1).sum())
    return wins_a, wins_b, int(len(res) - wins_a - wins_b), records


def mixed_arena_capable(agents) -> bool:
    """True for the reference's own pairing (run_tournament.py:44-48): one MinimaxAgent and one
    AlphaZeroAgent, in either order."""
    from .agent import AlphaZeroAgent
    from .minimax import MinimaxAgent
    if len(agents) != 2:
        return False
    kinds = {type(a) for a in agents}
    return kinds == {MinimaxAgent, AlphaZeroAgent} and all(
        a.num_simulations >= 2 for a in agents if isinstance(a, AlphaZeroAgent))


def mixed_arena(agents, num_games: int, keep_agent_sides: bool = False):
    """AlphaZeroAgent against MinimaxAgent, all games of the tournament advanced together: at every
    ply the positions where the minimax agent is to move go through one nnb_minimax call and those
    where the AlphaZero agent is to move through batched searches; moves are applied with the rule
    kernel.  Same protocol as `play_match` (agents[0] plays colour 1, game i is started by
    agents[i % 2], a missing move counts as a draw); returns the list of winners, one of
    0 (agents[0]), 1 (agents[1]) or None (draw) per game."""
    from .agent import AlphaZeroAgent
    if not mixed_arena_capable(agents):
        raise ValueError("mixed_arena needs one MinimaxAgent and one AlphaZeroAgent")
    az_seat = 0 if isinstance(agents[0], AlphaZeroAgent) else 1
    az, mm = agents[az_seat], agents[1 - az_seat]
    if az.load_model_flag and not az.model_loaded:
        az.load_model()
    n = int(num_games)
    chunk = min(max(n, 1), ARENA_CONCURRENT_GAMES)
    eng = az.engine(chunk)
    params = eng.params(az.num_simulations, az.c_puct, evaluator=az.evaluator, mode="compat")
    boards = np.zeros((n, 2), dtype=np.uint64)
    seat = (np.arange(n) % 2).astype(np.int64)              # whose turn it is; seat s plays colour s + 1
    live = np.ones(n, dtype=bool)
    winners = [None] * n
    for _ply in range(42):
        if not live.any():
            break
        action = np.full(n, -1, dtype=np.int8)
        idx = np.nonzero(live & (seat != az_seat))[0]
        if len(idx):
            col, _ = eng.minimax(boards[idx], (seat[idx] + 1).astype(np.int8), mm.depth)
            action[idx] = col
        idx = np.nonzero(live & (seat == az_seat))[0]
        for lo in range(0, len(idx), chunk):
            part = idx[lo:lo + chunk]
            movers = (seat[part] + 1).astype(np.int8)
            sides = np.full(len(part), 2 if az.current_player == 2 else 1, dtype=np.int8) if keep_agent_sides else movers
            _, act, _, _ = eng.search(boards[part], movers, params, side_players=sides)
            action[part] = act
        live &= action >= 0                                  # no move available: the game counts as a draw
        idx = np.nonzero(live)[0]
        if not len(idx):
            break
        new, _legal, status = eng.game_ops(boards[idx], (seat[idx] + 1).astype(np.int8), action[idx])
        boards[idx] = new
        for k, i in enumerate(idx):
            st = int(status[k])
            if not st & 1:                                   # column was full: not a valid location
                live[i] = False
            elif st & 6:                                       # somebody has four: it is the mover
                winners[i] = int(seat[i])
                live[i] = False
            elif st & 8:
                live[i] = False
        seat[idx] = 1 - seat[idx]
    return winners


def run_tournament(agents, num_games=100, results_dir=os.path.join("tournament", "results"), batched=None,
                   keep_agent_sides: bool = False):
    """run_tournament.py:12-42.  batched=None picks a batched engine arena when the pair allows it.
    keep_agent_sides: see play_match."""
    names = _names(agents)
    results = {n: 0 for n in names}
    results["draws"] = 0
    if batched is None:
        import torch
        batched = arena_capable(agents) or (mixed_arena_capable(agents) and torch.cuda.is_available())
    if batched and arena_capable(agents):
        wa, wb, dr, _ = arena(agents, num_games, keep_agent_sides=keep_agent_sides)
        results[names[0]], results[names[1]], results["draws"] = wa, wb, dr
    elif batched and mixed_arena_capable(agents):
        for w in mixed_arena(agents, num_games, keep_agent_sides=keep_agent_sides):
            results["draws" if w is None else names[w]] += 1
    else:
        game = ConnectFourGame()
        for i in range(num_games):
            results[play_match(agents, names, game, i % 2, keep_agent_sides)] += 1
    os.makedirs(results_dir, exist_ok=True)
    path = os.path.join(results_dir, "tournament_results.json")
    with open(path, "w") as f:
        json.dump(results, f, indent=4)
    print(f"Tournament completed. Results saved to {path}")
    return results


__all__ = ["run_tournament", "play_match", "arena", "arena_capable", "mixed_arena", "mixed_arena_capable"]

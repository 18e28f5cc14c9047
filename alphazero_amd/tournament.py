"""Agent-versus-agent arena with the interface of nnbattle/tournament/run_tournament.py:12-42.

Same protocol: `num_games` games, the starting agent alternates with the game index, the
first agent of the pair plays colour 1, a win is credited to the winner, anything else
(no legal move, full board) counts as a draw, and the tally is written as JSON to
tournament/results/tournament_results.json.  Agents are asked for a move with the game
object (what AlphaZeroAgent.select_move and MinimaxAgent.select_move both take) and may
return either a column or a (column, distribution) pair.  Tallies are keyed by class name,
with an index appended when two agents share a class.
"""
from __future__ import annotations

import json
import os

from .game import AI_PIECE, PLAYER_PIECE, ConnectFourGame


def _names(agents):
    base = [a.__class__.__name__ for a in agents]
    return [n if base.count(n) == 1 else f"{n}_{i}" for i, n in enumerate(base)]


def _column(move):
    if isinstance(move, (tuple, list)):
        move = move[0]
    return None if move is None else int(move)


def play_match(agents, names, game, starter: int):
    """One game; returns the winning agent's name or 'draws'."""
    game.reset()
    seat = starter
    while not game.is_terminal_node():
        agent = agents[seat]
        piece = PLAYER_PIECE if seat == 0 else AI_PIECE
        game.current_player = piece
        if hasattr(agent, "current_player"):
            agent.current_player = piece
        col = _column(agent.select_move(game))
        if col is None or not game.is_valid_location(col):
            return "draws"
        game.drop_piece(game.get_next_open_row(col), col, piece)
        if game.check_win(piece):
            return names[seat]
        seat = 1 - seat
    return "draws"


def run_tournament(agents, num_games=100, results_dir=os.path.join("tournament", "results")):
    names = _names(agents)
    results = {n: 0 for n in names}
    results["draws"] = 0
    game = ConnectFourGame()
    for i in range(num_games):
        results[play_match(agents, names, game, i % 2)] += 1
    os.makedirs(results_dir, exist_ok=True)
    path = os.path.join(results_dir, "tournament_results.json")
    with open(path, "w") as f:
        json.dump(results, f, indent=4)
    print(f"Tournament completed. Results saved to {path}")
    return results


__all__ = ["run_tournament", "play_match"]

"""Depth-limited alpha-beta opponent (CPU), interface of nnbattle/agents/minimax/agent_code.py.

Kept so arena games against the baseline agent still run; it is not accelerated.
"""
from __future__ import annotations

import math
import random

from .agent import Agent
from .game import AI_PIECE, PLAYER_PIECE, ConnectFourGame


class MinimaxAgent(Agent):
    def __init__(self, depth=4):
        self.depth = depth

    def select_move(self, game: ConnectFourGame):
        """Column chosen by alpha-beta search for the side to move in `game`."""
        me = game.current_player
        column, _ = self._search(game.copy(), self.depth, -math.inf, math.inf, True, me)
        return column

    def _search(self, game, depth, alpha, beta, maximizing, me):
        rival = PLAYER_PIECE if me == AI_PIECE else AI_PIECE
        moves = game.get_valid_locations()
        if game.check_win(me):
            return None, 1e13
        if game.check_win(rival):
            return None, -1e13
        if not moves:
            return None, 0
        if depth == 0:
            return None, game.score_position(me)
        best_col = random.choice(moves)
        best = -math.inf if maximizing else math.inf
        for col in moves:
            child = game.copy()
            child.current_player = me if maximizing else rival
            child.make_move(col)
            _, score = self._search(child, depth - 1, alpha, beta, not maximizing, me)
            if maximizing and score > best:
                best, best_col = score, col
                alpha = max(alpha, best)
            elif not maximizing and score < best:
                best, best_col = score, col
                beta = min(beta, best)
            if alpha >= beta:
                break
        return best_col, best


__all__ = ["MinimaxAgent"]

"""nnbattle.game.connect_four_game -> alphazero_amd.game."""
from alphazero_amd.game import (AI_PIECE, COLUMN_COUNT, EMPTY, PLAYER_PIECE, ROW_COUNT, WINDOW_LENGTH,
                                ConnectFourGame)

__all__ = ["ConnectFourGame", "PLAYER_PIECE", "AI_PIECE", "EMPTY", "ROW_COUNT", "COLUMN_COUNT", "WINDOW_LENGTH"]

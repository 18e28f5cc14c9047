"""Host-side Connect Four game object with the reference's public API.

`ConnectFourGame` keeps the reference's contract (nnbattle/game/connect_four_game.py:19-304):
`board` is a public, externally assignable `int8[6,7]` numpy array whose row 0 is the
top row, `current_player` is 1 or 2, and every method keeps its name, arguments and
return values.  Because callers may overwrite `board` at any time, the array stays the
single source of truth; rule queries convert it to the engine's two-word bitboard form
(bit = col*7 + (5 - row)) and answer with a few shifts instead of scanning 69 windows.
The same bitboards are what the device engine consumes (`colour_boards()`).
"""
from __future__ import annotations

import logging

import numpy as np

logger = logging.getLogger(__name__)

PLAYER_PIECE = 1
AI_PIECE = 2
EMPTY = 0
ROW_COUNT = 6
COLUMN_COUNT = 7
WINDOW_LENGTH = 4

_BIT = np.array([[1 << (c * 7 + (ROW_COUNT - 1 - r)) for c in range(COLUMN_COUNT)]
                 for r in range(ROW_COUNT)], dtype=np.int64)
_FULL = int(_BIT.sum())


def _windows():
    """The 69 windows in the order get_win_positions reports them
    (connect_four_game.py:168-191): horizontal, vertical, (1,1) diagonals, (-1,1) diagonals."""
    spec = []
    spec += [((r, c), (0, 1)) for r in range(ROW_COUNT) for c in range(COLUMN_COUNT - 3)]
    spec += [((r, c), (1, 0)) for c in range(COLUMN_COUNT) for r in range(ROW_COUNT - 3)]
    spec += [((r, c), (1, 1)) for r in range(ROW_COUNT - 3) for c in range(COLUMN_COUNT - 3)]
    spec += [((r, c), (-1, 1)) for r in range(3, ROW_COUNT) for c in range(COLUMN_COUNT - 3)]
    out = []
    for (r, c), (dr, dc) in spec:
        cells = [(r + i * dr, c + i * dc) for i in range(WINDOW_LENGTH)]
        mask = 0
        for rr, cc in cells:
            mask |= int(_BIT[rr][cc])
        out.append((((r, c), (dr, dc)), mask, cells))
    return out


_WINDOWS = _windows()


def board_to_bits(board: np.ndarray):
    """int8[6,7] board -> (colour-1 bits, colour-2 bits)."""
    return int(_BIT[board == PLAYER_PIECE].sum()), int(_BIT[board == AI_PIECE].sum())


def bits_to_board(b1: int, b2: int) -> np.ndarray:
    board = np.zeros((ROW_COUNT, COLUMN_COUNT), dtype=np.int8)
    board[(_BIT & b1) != 0] = PLAYER_PIECE
    board[(_BIT & b2) != 0] = AI_PIECE
    return board


def has_four(bits: int) -> bool:
    for shift in (1, 7, 6, 8):
        m = bits & (bits >> shift)
        if m & (m >> (2 * shift)):
            return True
    return False


class ConnectFourGame:
    def __init__(self):
        self.board = np.zeros((ROW_COUNT, COLUMN_COUNT), dtype=np.int8)
        self.current_player = PLAYER_PIECE

    # -- state ------------------------------------------------------------
    def copy(self):
        twin = ConnectFourGame()
        twin.board = self.board.copy()
        twin.current_player = self.current_player
        return twin

    def reset(self):
        self.board = np.zeros((ROW_COUNT, COLUMN_COUNT), dtype=np.int8)
        self.current_player = PLAYER_PIECE
        return self.board.copy()

    def get_board_state(self):
        return self.board.copy()

    def get_state(self):
        return self.board.copy()

    def colour_boards(self):
        """(colour-1 bits, colour-2 bits) of the current board, the engine's input form."""
        return board_to_bits(self.board)

    # -- moves ------------------------------------------------------------
    def is_valid_location(self, col):
        assert isinstance(self.board, np.ndarray), f"self.board is not a NumPy array but {type(self.board)}"
        return self.board[0][col] == EMPTY

    def get_valid_locations(self):
        return [c for c in range(COLUMN_COUNT) if self.is_valid_location(c)]

    def get_next_open_row(self, col):
        empties = np.flatnonzero(self.board[:, col] == EMPTY)
        return int(empties[-1]) if empties.size else None

    def drop_piece(self, row, col, piece):
        self.board[row][col] = piece

    def make_move(self, action):
        """Drop the mover's stone in column `action`; a full column is refused without
        switching the mover (connect_four_game.py:38-52)."""
        if not self.is_valid_location(action):
            return False
        row = self.get_next_open_row(action)
        self.board[row][action] = self.current_player
        self.current_player = AI_PIECE if self.current_player == PLAYER_PIECE else PLAYER_PIECE
        return True

    def step(self, action):
        """(board, reward, done, info); an illegal move ends the episode with -10
        (connect_four_game.py:54-70)."""
        if not self.make_move(action):
            return self.board.copy(), -10.0, True, {"invalid_move": True}
        return self.board.copy(), self.get_reward(), self.is_terminal(), {}

    # -- outcome ----------------------------------------------------------
    def get_win_positions(self, player):
        bits = int(_BIT[self.board == player].sum())
        return [tag for tag, mask, _ in _WINDOWS if bits & mask == mask]

    def check_win(self, player):
        return has_four(int(_BIT[self.board == player].sum()))

    def is_board_full(self):
        return not (self.board == EMPTY).any()

    def is_terminal(self):
        b1, b2 = board_to_bits(self.board)
        return has_four(b1) or has_four(b2) or self.is_board_full()

    def is_terminal_node(self):
        return self.is_terminal()

    def get_reward(self):
        """+1 if colour 2 has four, -1 if colour 1 has, else 0 (connect_four_game.py:80-91)."""
        b1, b2 = board_to_bits(self.board)
        if has_four(b2):
            return 1.0
        if has_four(b1):
            return -1.0
        return 0.0

    def get_winner(self):
        b1, b2 = board_to_bits(self.board)
        if has_four(b2):
            return AI_PIECE
        if has_four(b1):
            return PLAYER_PIECE
        return EMPTY

    def get_result(self):
        """1 / -1 / 0 in colour 1's frame once the game is over, else None
        (connect_four_game.py:280-296)."""
        b1, b2 = board_to_bits(self.board)
        if has_four(b1):
            return 1
        if has_four(b2):
            return -1
        return 0 if self.is_board_full() else None

    # -- heuristic used by the minimax opponent ------------------------------
    def evaluate_window(self, window, piece):
        """Score of one window of four cells (connect_four_game.py:257-278)."""
        window = [int(v) for v in window]
        rival = PLAYER_PIECE if piece == AI_PIECE else AI_PIECE
        mine, free, theirs = window.count(piece), window.count(EMPTY), window.count(rival)
        score = 0
        if mine == 4:
            score += 100
        elif mine == 3 and free == 1:
            score += 5
        elif mine == 2 and free == 2:
            score += 2
        if theirs == 3 and free == 1:
            score -= 4
        return score

    def score_position(self, piece):
        """Centre-column bonus plus the window scores over all 69 windows
        (connect_four_game.py:215-255)."""
        score = 3 * int((self.board[:, COLUMN_COUNT // 2] == piece).sum())
        for _, _, cells in _WINDOWS:
            score += self.evaluate_window([self.board[r][c] for r, c in cells], piece)
        return score

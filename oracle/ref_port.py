"""CPU restatement of the nnbattle Connect Four + MCTS self-play path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this file:
it exists so tests, the smoke check and the benchmark's CPU-baseline leg have an
independent statement of what the reference computes.  It is written to follow
the reference's cost structure (numpy int8 board, per-window win scan, one
batch-1 network call per simulation) so that timing it is a fair stand-in for
timing the reference itself.

Every function cites the reference lines (under /root/reference) it restates.
Parity is pinned: tests/test_oracle_golden.py checks this module against
vectors captured from the imported reference by oracle/capture_golden.py.
"""
from __future__ import annotations

import math

import numpy as np

ROWS = 6            # connect_four_game.py:15
COLS = 7            # connect_four_game.py:16
P1 = 1              # connect_four_game.py:10  (PLAYER_PIECE)
P2 = 2              # connect_four_game.py:13  (AI_PIECE)

MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------
# Game rules (connect_four_game.py)
# --------------------------------------------------------------------------
def new_board() -> np.ndarray:
    """connect_four_game.py:21 — empty int8[6,7], row 0 is the top row."""
    return np.zeros((ROWS, COLS), dtype=np.int8)


def other(player: int) -> int:
    """connect_four_game.py:50 — colour flip."""
    return P2 if player == P1 else P1


def is_valid_location(board: np.ndarray, col: int) -> bool:
    """connect_four_game.py:117-125 — a column is open iff its top cell is empty."""
    return board[0][col] == 0


def valid_locations(board: np.ndarray) -> list:
    """connect_four_game.py:109-115 — ascending list of open columns."""
    return [c for c in range(COLS) if board[0][c] == 0]


def make_move(board: np.ndarray, player: int, col: int):
    """connect_four_game.py:38-52 — drop `player` in `col` (lowest empty row).

    Returns (ok, next_player).  On a full column nothing changes and the player
    is NOT flipped (pinned by game/tests/test_connect_four_game.py:43-51).
    """
    if board[0][col] == 0:
        for row in range(ROWS - 1, -1, -1):
            if board[row][col] == 0:
                board[row][col] = player
                return True, other(player)
    return False, player


def win_positions(board: np.ndarray, player: int) -> list:
    """connect_four_game.py:160-192 — all 69 windows of four, same scan order."""
    wins = []
    for row in range(ROWS):                      # horizontal, :169-173
        for col in range(COLS - 3):
            if np.all(board[row, col:col + 4] == player):
                wins.append(((row, col), (0, 1)))
    for col in range(COLS):                      # vertical, :175-179
        for row in range(ROWS - 3):
            if np.all(board[row:row + 4, col] == player):
                wins.append(((row, col), (1, 0)))
    for row in range(ROWS - 3):                  # diagonal down-right, :181-185
        for col in range(COLS - 3):
            if all(board[row + i][col + i] == player for i in range(4)):
                wins.append(((row, col), (1, 1)))
    for row in range(3, ROWS):                   # diagonal up-right, :187-191
        for col in range(COLS - 3):
            if all(board[row - i][col + i] == player for i in range(4)):
                wins.append(((row, col), (-1, 1)))
    return wins


def check_win(board: np.ndarray, player: int) -> bool:
    """connect_four_game.py:148-158."""
    return len(win_positions(board, player)) > 0


def is_board_full(board: np.ndarray) -> bool:
    """connect_four_game.py:194-200."""
    return not (board == 0).any()


def is_terminal(board: np.ndarray) -> bool:
    """connect_four_game.py:72-78."""
    return check_win(board, P1) or check_win(board, P2) or is_board_full(board)


def get_reward(board: np.ndarray) -> float:
    """connect_four_game.py:80-91 — absolute frame: +1 colour 2 wins, -1 colour 1."""
    if check_win(board, P2):
        return 1.0
    if check_win(board, P1):
        return -1.0
    return 0.0


def get_winner(board: np.ndarray) -> int:
    """connect_four_game.py:202-213."""
    if check_win(board, P2):
        return P2
    if check_win(board, P1):
        return P1
    return 0


def get_result(board: np.ndarray):
    """connect_four_game.py:280-296 — player-1 frame, None while running."""
    if is_terminal(board):
        if check_win(board, 1):
            return 1
        if check_win(board, 2):
            return -1
        return 0
    return None


# --------------------------------------------------------------------------
# Encodings shared by fixtures, oracles and the device code
# --------------------------------------------------------------------------
def board_to_str(board: np.ndarray) -> str:
    """42 characters, row-major from the top row (fixture encoding)."""
    return "".join(str(int(v)) for v in board.reshape(-1))


def board_from_str(s: str) -> np.ndarray:
    return np.array([int(ch) for ch in s], dtype=np.int8).reshape(ROWS, COLS)


def bitboards(board: np.ndarray):
    """(colour-1 bits, colour-2 bits); bit index = col*7 + (5 - row)."""
    b1 = b2 = 0
    for r in range(ROWS):
        for c in range(COLS):
            v = int(board[r][c])
            if v:
                bit = 1 << (c * 7 + (ROWS - 1 - r))
                if v == P1:
                    b1 |= bit
                else:
                    b2 |= bit
    return b1, b2


def planes(board: np.ndarray, me: int) -> np.ndarray:
    """agent_code.py:84-88 — float32[2,6,7]: [board==me, board==opponent]."""
    opp = other(me)
    return np.stack([(board == me), (board == opp)]).astype(np.float32)


def mix64(me_bits: int, opp_bits: int) -> int:
    """Position hash used by the fixed evaluators (splitmix64 finaliser)."""
    x = (me_bits ^ ((opp_bits * 0x9E3779B97F4A7C15) & MASK64)) & MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x


def hash_value_q8(me_bits: int, opp_bits: int) -> float:
    """Value in [-1, 1): a multiple of 1/256, so float64 sums are exact."""
    return ((mix64(me_bits, opp_bits) >> 55) - 256) / 256.0


def hash_value_f32(me_bits: int, opp_bits: int) -> float:
    """Value in [-1, 1): 24-bit mantissa scaled by 2^-e, e in 0..7 (exact in float32)."""
    x = mix64(me_bits, opp_bits)
    return float(np.float32(((x >> 40) / 8388608.0 - 1.0) / float(1 << ((x >> 37) & 7))))


# --------------------------------------------------------------------------
# Evaluators: callable(board, me) -> value (Python float, already widened)
# --------------------------------------------------------------------------
def eval_const(board, me):
    """Uniform policy, value 0 (policy is discarded by the search anyway)."""
    return 0.0


def _me_opp_bits(board, me):
    b1, b2 = bitboards(board)
    return (b1, b2) if me == P1 else (b2, b1)


def eval_hash_q8(board, me):
    return hash_value_q8(*_me_opp_bits(board, me))


def eval_hash_f32(board, me):
    return hash_value_f32(*_me_opp_bits(board, me))


EVALUATORS = {"const": eval_const, "hash_q8": eval_hash_q8, "hash_f32": eval_hash_f32}


def net_forward(sd: dict, x):
    """network.py:43-50 — functional Connect4Net forward from a state dict."""
    import torch
    import torch.nn.functional as F
    x = F.relu(F.conv2d(x, sd["conv_layers.0.weight"], sd["conv_layers.0.bias"], padding=2))
    x = F.relu(F.conv2d(x, sd["conv_layers.2.weight"], sd["conv_layers.2.bias"], padding=2))
    x = F.relu(F.conv2d(x, sd["conv_layers.4.weight"], sd["conv_layers.4.bias"], padding=2))
    x = x.reshape(x.size(0), -1)
    x = F.relu(F.linear(x, sd["fc_layers.0.weight"], sd["fc_layers.0.bias"]))
    log_policy = F.log_softmax(F.linear(x, sd["policy_head.weight"], sd["policy_head.bias"]), dim=1)
    value = torch.tanh(F.linear(x, sd["value_head.weight"], sd["value_head.bias"]))
    return log_policy, value


def seeded_state_dict(seed: int) -> dict:
    """Random-init weights of network.py:28-41 under torch.manual_seed(seed).

    The module is rebuilt layer by layer in the reference's construction order,
    so the generator is consumed exactly as Connect4Net.__init__ consumes it.
    """
    import torch
    import torch.nn as nn
    torch.manual_seed(seed)
    convs = nn.Sequential(
        nn.Conv2d(2, 128, kernel_size=4, stride=1, padding=2), nn.ReLU(),
        nn.Conv2d(128, 128, kernel_size=4, stride=1, padding=2), nn.ReLU(),
        nn.Conv2d(128, 128, kernel_size=4, stride=1, padding=2), nn.ReLU())
    fc = nn.Sequential(nn.Linear(128 * 9 * 10, 1024), nn.ReLU())
    ph = nn.Linear(1024, 7)
    vh = nn.Linear(1024, 1)
    sd = {}
    for prefix, mod in (("conv_layers", convs), ("fc_layers", fc)):
        for k, v in mod.state_dict().items():
            sd[f"{prefix}.{k}"] = v.detach().clone()
    for prefix, mod in (("policy_head", ph), ("value_head", vh)):
        for k, v in mod.state_dict().items():
            sd[f"{prefix}.{k}"] = v.detach().clone()
    return sd


def make_net_evaluator(sd: dict):
    """agent_code.py:195-199,213 — batch-1 forward, value.item() widened to double."""
    import torch

    def _eval(board, me):
        x = torch.from_numpy(planes(board, me)).unsqueeze(0)
        with torch.no_grad():
            _, value = net_forward(sd, x)
        return value.item()
    return _eval


def masked_priors(log_policy_row: np.ndarray, legal: list) -> np.ndarray:
    """agent_code.py:199-208 — exp, zero illegal, renormalise (float64), uniform fallback."""
    policy = np.exp(log_policy_row.astype(np.float32))       # torch.exp on float32
    filtered = np.zeros(COLS)
    filtered[legal] = policy[legal]
    if filtered.sum() > 0:
        filtered /= filtered.sum()
    else:
        filtered[legal] = 1.0 / len(legal)
    return filtered


# --------------------------------------------------------------------------
# Search (agent_code.py:144-234 + mcts.py:13-44), flat arrays instead of objects
# --------------------------------------------------------------------------
def compat_search(board: np.ndarray, player: int, num_simulations: int,
                  c_puct: float, evaluator, stats: dict | None = None, side: int | None = None):
    """One reference search from (board, player to move).

    `player` is env.current_player (agent_code.py:167), the colour placed during
    descent; `side` is agent.current_player (agent_code.py:85-87), the colour the
    leaf planes are encoded from.  The reference reads the two independently;
    its self-play keeps them equal (train_alpha_zero.py:38,45), which is the
    default here (side=None).

    Returns (selected_action, visits[7]) where visits are the root children's
    visit counts (0 for columns without a child).  Quirks kept on purpose:
      * the root starts with visits = 1 and is a leaf (agent_code.py:159-160);
      * descent places the ROOT mover's colour at every depth, because the
        player is flipped by make_move (connect_four_game.py:50) and flipped
        back at agent_code.py:176;
      * the leaf is encoded from the agent's side, `side`, at every depth (agent_code.py:85);
      * terminal reward is absolute (+1 colour 2 / -1 colour 1) and the backup
        does not negate it (mcts.py:40-44);
      * child priors are discarded (mcts.py:35-38);
      * ties go to the lowest column (strict '>' at mcts.py:30, first max at
        agent_code.py:223).
    """
    if side is None:
        side = player
    visits = [1]          # agent_code.py:160
    value = [0.0]
    parent = [-1]
    action = [-1]
    first_child = [-1]
    n_child = [0]
    depth_sum = 0
    net_calls = 0

    for _ in range(num_simulations):                     # agent_code.py:162
        node = 0
        b = board.copy()                                 # agent_code.py:164-167
        mover = player
        depth = 0
        # selection, agent_code.py:170-176
        while n_child[node] > 0 and not is_terminal(b):
            best_score = -float("inf")                   # mcts.py:26
            best = -1
            fc = first_child[node]
            log_term = math.log(visits[node] + 1)        # mcts.py:29
            for ch in range(fc, fc + n_child[node]):
                score = (value[ch] / (visits[ch] + 1)) + c_puct * math.sqrt(log_term / (visits[ch] + 1))
                if score > best_score:
                    best_score = score
                    best = ch
            node = best
            _, mover = make_move(b, mover, action[node])
            mover = other(mover)                         # agent_code.py:176
            depth += 1
        depth_sum += depth

        if is_terminal(b):                               # agent_code.py:182-186
            reward = get_reward(b)
        else:
            legal = valid_locations(b)                   # agent_code.py:189
            if not legal:                                # agent_code.py:190-193
                reward = 0
            else:
                reward = evaluator(b, side)              # agent_code.py:195-199,213
                net_calls += 1
                first_child[node] = len(visits)          # mcts.py:35-38
                n_child[node] = len(legal)
                for a in legal:
                    visits.append(0)
                    value.append(0.0)
                    parent.append(node)
                    action.append(a)
                    first_child.append(-1)
                    n_child.append(0)
        n = node                                         # mcts.py:40-44
        while n >= 0:
            visits[n] += 1
            value[n] += reward
            n = parent[n]

    out = [0] * COLS
    if n_child[0] == 0:                                  # agent_code.py:219-221
        return None, out
    best_a, best_v = -1, -1
    for ch in range(first_child[0], first_child[0] + n_child[0]):
        out[action[ch]] = visits[ch]
        if visits[ch] > best_v:                          # first max, agent_code.py:223
            best_v = visits[ch]
            best_a = action[ch]
    if stats is not None:
        stats["nodes"] = len(visits)
        stats["depth_sum"] = stats.get("depth_sum", 0) + depth_sum
        stats["net_calls"] = stats.get("net_calls", 0) + net_calls
        stats["root_visits"] = visits[0]
        stats["root_value"] = value[0]
    return best_a, out


def self_play_game(num_simulations: int, c_puct: float, evaluator, stats: dict | None = None,
                   max_plies: int | None = None):
    """train_alpha_zero.py:36-48 — one game, same agent on both sides.

    Returns (moves, per-ply root visit vectors, result in player-1 frame).
    """
    board = new_board()
    player = P1
    moves, visit_log = [], []
    while not is_terminal(board):
        a, v = compat_search(board, player, num_simulations, c_puct, evaluator, stats)
        moves.append(a)
        visit_log.append(v)
        _, player = make_move(board, player, a)
        if max_plies is not None and len(moves) >= max_plies:
            break
    return moves, visit_log, get_result(board), board

"""AlphaZero agent with the reference's class API, searching on the MI355X engine.

Mirrors nnbattle/agents/alphazero/agent_code.py:23-274, utils.py:21-86 and
base_agent.py:6-12: same constructor, attributes, method names and return shapes.
What changed is where the work happens: `mcts_simulate` no longer walks Python
objects and calls the network once per simulation; it hands the root position to
`libnnb.so`, which runs the whole search on the GPU (csrc/tree_kernels.hip,
csrc/net_kernels.hip).  There is no CPU search path: without a GPU, or without the
built library, `select_move` raises.
"""
from __future__ import annotations

import copy
import logging
import os
import time
from abc import ABC, abstractmethod

import numpy as np
import torch

from .game import ConnectFourGame, board_to_bits
from .network import Connect4Net

logger = logging.getLogger(__name__)

MODEL_PATH = "nnbattle/agents/alphazero/model/alphazero_model_final.pth"   # utils.py:31
DEFAULT_PRECISION = "f32s"


class Agent(ABC):
    """base_agent.py:6-12."""

    @abstractmethod
    def select_move(self, board):
        """Return the column (0-6) the agent wants to play."""


def deepcopy_env(env):
    """utils.py:21-28."""
    return copy.deepcopy(env)


def preprocess_board(board, player):
    """utils.py:11-19 (unused by the live path, kept for import compatibility)."""
    return copy.deepcopy(board) * player


class AlphaZeroAgent(Agent):
    logger = logging.getLogger(__name__)

    def __init__(self, action_dim, state_dim=2, use_gpu=False, num_simulations=800, c_puct=1.4,
                 load_model=True):
        super().__init__()
        self.state_dim = state_dim
        self.action_dim = action_dim
        # where the model object and returned tensors live (agent_code.py:39); the search
        # itself always runs on the GPU engine
        self.device = torch.device("cuda" if use_gpu and torch.cuda.is_available() else "cpu")
        self.model_loaded = False
        self.load_model_flag = load_model
        self.num_simulations = num_simulations
        self.c_puct = c_puct
        self.current_player = 1
        self.memory = []
        self.model = Connect4Net(state_dim, action_dim).to(self.device)
        self.logger = self.__class__.logger
        self.evaluator = "net"          # fixed evaluators ("const", "hash_q8", "hash_f32") for parity runs
        self.eval_cache = False         # True: a game does not send one position to the network twice (same moves)
        # arithmetic of the network's conv / linear layers on the device (engine.PRECISIONS): "f32s" is
        # float32-faithful (the reference computes in float32, network.py:43-50), "f16" rounds the
        # operands to half and runs three times as fast, "f32" is the exact-FMA matrix path
        self.precision = DEFAULT_PRECISION
        self._engine = None
        self._engine_key = None
        self._weights_stamp = None
        logger.info("Using device: %s", self.device)
        if load_model:
            load_agent_model(self)

    # ------------------------------------------------------------------ engine plumbing
    def _weights_fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.state_dict().values())

    def engine(self, max_games: int = 1, max_sims: int = 0):
        """The device engine sized for this agent (created on first use, re-created if the
        simulation count or the requested game count grows)."""
        from .engine import Engine
        key = (max(int(max_games), 1), max(int(self.num_simulations), int(max_sims), 1), bool(self.eval_cache),
               str(self.precision))
        if (self._engine is None or self._engine_key[0] < key[0] or self._engine_key[1] < key[1]
                or self._engine_key[2:] != key[2:]):
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(max_games=key[0], max_sims=key[1], eval_cache_slots=2048 if key[2] else 0,
                                  precision=key[3])
            self._engine_key = key
            self._weights_stamp = None
        if self.evaluator == "net":
            stamp = self._weights_fingerprint()
            if stamp != self._weights_stamp:
                self._engine.load_state_dict(self.model.state_dict())
                self._weights_stamp = stamp
        return self._engine

    def log_gpu_stats(self):
        if torch.cuda.is_available():
            logger.info("GPU: %s", torch.cuda.get_device_name())
            logger.info("GPU Memory - Allocated: %.2f GB", torch.cuda.memory_allocated() / 1e9)
            logger.info("GPU Memory - Reserved: %.2f GB", torch.cuda.memory_reserved() / 1e9)
            logger.info("GPU Memory - Peak: %.2f GB", torch.cuda.max_memory_allocated() / 1e9)
            torch.cuda.reset_peak_memory_stats()

    # ------------------------------------------------------------------ reference API
    def preprocess(self, board):
        """[board == mover, board == opponent] as float32 [2,6,7] on self.device
        (agent_code.py:76-89); the mover is self.current_player."""
        board = np.asarray(board)
        rival = 2 if self.current_player == 1 else 1
        planes = np.stack([board == self.current_player, board == rival]).astype(np.float32)
        return torch.from_numpy(planes).to(self.device)

    def load_model(self):
        logger.warning("load_model method is deprecated. Use load_agent_model from utils.py instead.")
        load_agent_model(self)

    def save_model(self, path=MODEL_PATH):
        save_agent_model(self, path)

    def select_move(self, game: ConnectFourGame):
        """(column, visit distribution FloatTensor[7]) or None (agent_code.py:100-120)."""
        if self.load_model_flag and not self.model_loaded:
            self.load_model()
        t0 = time.time()
        selected_action, _actions, action_probs = self.act(game.board, game, self.num_simulations)
        logger.info("Time taken for select_move: %.4f seconds", time.time() - t0)
        if selected_action is None:
            logger.warning("No possible actions available.")
            return None
        return selected_action, action_probs

    def act(self, state, env, num_simulations=800):
        """agent_code.py:122-142."""
        selected_action, actions, probabilities = self.mcts_simulate(state, env, num_simulations)
        if selected_action is None:
            logger.warning("No valid action selected by MCTS.")
            return None, [], []
        return selected_action, actions, probabilities

    def mcts_simulate(self, state, env, num_simulations=800):
        """One search from `env` (agent_code.py:144-234), run on the GPU.

        Returns (selected column, columns that got a child, visit distribution) exactly as
        the reference does: the most visited root child with the lowest column on ties, and
        visits/total as a FloatTensor[7] on self.device.  Like the reference, the colour
        placed during descent is `env.current_player` (agent_code.py:167,176) and the side
        the network looks from is `self.current_player` (agent_code.py:85-87); the two are
        handed to the engine separately.  self_play keeps them equal
        (train_alpha_zero.py:38,45); run_tournament.py:21-22 never updates the agent's.
        """
        eng = self.engine(1, num_simulations)
        b1, b2 = board_to_bits(np.asarray(env.board))
        params = eng.params(num_simulations, self.c_puct, evaluator=self.evaluator, mode="compat",
                            eval_cache=self.eval_cache)
        visits, action, _rootv, _stats = eng.search(np.array([[b1, b2]], dtype=np.uint64),
                                                    np.array([env.current_player], dtype=np.int8), params,
                                                    side_players=np.array([2 if self.current_player == 2 else 1],
                                                                          dtype=np.int8))
        if int(action[0]) < 0:
            logger.error("No actions available after MCTS simulations.")
            return None, [], []
        legal = [c for c in range(self.action_dim) if env.board[0][c] == 0]
        total = int(visits[0].sum())
        action_probs = np.zeros(self.action_dim)
        for c in legal:
            action_probs[c] = int(visits[0][c]) / total     # ZeroDivisionError at 1 simulation, as upstream
        return int(action[0]), legal, torch.FloatTensor(action_probs).to(self.device)

    def self_play(self):
        """One self-play game appended to self.memory as (board, visit distribution, value)
        samples (agent_code.py:236-267; the reference version reads self.env, which it never
        sets - here a fresh game is used)."""
        env = ConnectFourGame()
        state = env.reset()
        game_data = []
        player = 1
        done = False
        while not done:
            self.current_player = env.current_player
            picked = self.select_move(env)
            if picked is None:
                logger.error("Agent failed to select a valid action during self-play.")
                break
            action, probs = picked
            game_data.append((state.copy(), probs, player))
            state, _reward, done, _ = env.step(action)
            player = -player
        winner = env.get_winner()
        winner_sign = {0: 0, 1: 1, 2: -1}[winner]
        for s, p, who in game_data:
            self.memory.append((s, p, 0 if winner_sign == 0 else (1 if winner_sign == who else -1)))

    def perform_training(self):
        from .selfplay import train_alphazero
        train_alphazero(time_limit=3600, num_self_play_games=1000, use_gpu=self.device.type == "cuda",
                        load_model=self.model_loaded)


def load_agent_model(agent: "AlphaZeroAgent"):
    """utils.py:33-51 - load MODEL_PATH into agent.model, FileNotFoundError if absent."""
    if not os.path.exists(MODEL_PATH):
        logging.getLogger(__name__).error("Model path %s does not exist.", MODEL_PATH)
        raise FileNotFoundError(f"Model path {MODEL_PATH} does not exist.")
    agent.model.load_state_dict(torch.load(MODEL_PATH, map_location=agent.device, weights_only=True))
    agent.model_loaded = True


def save_agent_model(agent: "AlphaZeroAgent", path: str = MODEL_PATH):
    """utils.py:53-61."""
    torch.save(agent.model.state_dict(), path)
    logging.getLogger(__name__).info("Model saved to %s.", path)


def initialize_agent(action_dim=7, state_dim=2, use_gpu=False, num_simulations=800, c_puct=1.4,
                     load_model=True):
    """utils.py:63-86."""
    return AlphaZeroAgent(action_dim=action_dim, state_dim=state_dim, use_gpu=use_gpu,
                          num_simulations=num_simulations, c_puct=c_puct, load_model=load_model)


__all__ = ["Agent", "AlphaZeroAgent", "MODEL_PATH", "deepcopy_env", "initialize_agent", "load_agent_model",
           "preprocess_board", "save_agent_model"]

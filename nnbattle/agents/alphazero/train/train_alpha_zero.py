"""nnbattle.agents.alphazero.train.train_alpha_zero -> alphazero_amd.selfplay."""
from alphazero_amd.selfplay import self_play, train_alphazero

__all__ = ["self_play", "train_alphazero"]

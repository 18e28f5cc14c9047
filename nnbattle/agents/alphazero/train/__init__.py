from .train_alpha_zero import self_play, train_alphazero

__all__ = ["self_play", "train_alphazero"]

from .alphazero import AlphaZeroAgent
from .minimax import MinimaxAgent

__all__ = ["AlphaZeroAgent", "MinimaxAgent"]

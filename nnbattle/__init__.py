"""Import-path compatibility layer: the nnbattle names resolve to the MI355X engine's
host classes in alphazero_amd, so code written against the reference keeps its imports."""
from .agents import AlphaZeroAgent, MinimaxAgent
from .game import ConnectFourGame
from .tournament import run_tournament

__all__ = ["AlphaZeroAgent", "MinimaxAgent", "ConnectFourGame", "run_tournament"]

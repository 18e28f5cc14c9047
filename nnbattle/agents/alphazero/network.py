"""nnbattle.agents.alphazero.network -> alphazero_amd.network."""
from alphazero_amd.network import Connect4Net

__all__ = ["Connect4Net"]

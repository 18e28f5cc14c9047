"""nnbattle.agents.alphazero.mcts -> alphazero_amd.mcts."""
from alphazero_amd.mcts import MCTSNode

__all__ = ["MCTSNode"]

"""nnbattle.agents.minimax.agent_code -> alphazero_amd.minimax."""
from alphazero_amd.minimax import MinimaxAgent

__all__ = ["MinimaxAgent"]

"""nnbattle.agents.alphazero.agent_code -> alphazero_amd.agent."""
from alphazero_amd.agent import AlphaZeroAgent

__all__ = ["AlphaZeroAgent"]

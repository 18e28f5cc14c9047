"""nnbattle.agents.base_agent -> alphazero_amd.agent."""
from alphazero_amd.agent import Agent

__all__ = ["Agent"]

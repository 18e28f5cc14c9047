from .agent_code import AlphaZeroAgent
from .utils import initialize_agent, load_agent_model, save_agent_model

__all__ = ["AlphaZeroAgent", "initialize_agent", "load_agent_model", "save_agent_model"]

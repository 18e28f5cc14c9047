"""nnbattle.agents.alphazero.utils -> alphazero_amd.agent."""
from alphazero_amd.agent import (MODEL_PATH, deepcopy_env, initialize_agent, load_agent_model,
                                 preprocess_board, save_agent_model)

__all__ = ["MODEL_PATH", "deepcopy_env", "initialize_agent", "load_agent_model", "preprocess_board",
           "save_agent_model"]

from .agent_code import MinimaxAgent

__all__ = ["MinimaxAgent"]

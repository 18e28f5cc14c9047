from .run_tournament import run_tournament

__all__ = ["run_tournament"]

"""nnbattle.tournament.run_tournament -> alphazero_amd.tournament."""
from alphazero_amd.tournament import run_tournament

__all__ = ["run_tournament"]

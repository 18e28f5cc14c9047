"""nnbattle.tournament.run_tournament -> alphazero_amd.tournament."""
from alphazero_amd.tournament import run_tournament

__all__ = ["run_tournament"]

if __name__ == "__main__":          # the reference's own pairing, run_tournament.py:44-48
    import os

    from alphazero_amd.agent import MODEL_PATH, initialize_agent
    from alphazero_amd.minimax import MinimaxAgent

    agent1 = MinimaxAgent(depth=4)
    agent2 = initialize_agent(action_dim=7, load_model=os.path.exists(MODEL_PATH))
    run_tournament([agent1, agent2], num_games=100)

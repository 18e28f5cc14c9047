from .connect_four_game import ConnectFourGame

__all__ = ["ConnectFourGame"]

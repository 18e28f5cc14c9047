"""Policy-value network container with the reference's parameter names.

The engine evaluates the network with its own HIP kernels; this module exists so
that `agent.model` is still an `nn.Module` whose `state_dict()` / `load_state_dict()`
use the keys of nnbattle/agents/alphazero/network.py:28-41 (checkpoints written by
the reference load unchanged), and so CPU callers can still run a forward pass.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F

TRUNK_CHANNELS = 128
FC_IN = TRUNK_CHANNELS * 9 * 10      # three pad-2 4x4 convolutions grow 6x7 to 9x10
FC_OUT = 1024


class Connect4Net(nn.Module):
    """[B,2,6,7] -> (log_policy [B,action_dim], value [B,1]); network.py:25-50."""

    def __init__(self, state_dim: int = 2, action_dim: int = 7):
        super().__init__()
        layers = []
        c_in = state_dim
        for _ in range(3):          # indices 0, 2, 4 hold the convolutions, as in the reference
            layers += [nn.Conv2d(c_in, TRUNK_CHANNELS, kernel_size=4, stride=1, padding=2), nn.ReLU()]
            c_in = TRUNK_CHANNELS
        self.conv_layers = nn.Sequential(*layers)
        self.fc_layers = nn.Sequential(nn.Linear(FC_IN, FC_OUT), nn.ReLU())
        self.policy_head = nn.Linear(FC_OUT, action_dim)
        self.value_head = nn.Linear(FC_OUT, 1)

    def forward(self, x: torch.Tensor):
        if x.shape[1] != 2:
            raise AssertionError(f"Expected input with 2 channels, got {x.shape[1]} channels.")
        h = self.conv_layers(x).flatten(1)
        h = self.fc_layers(h)
        return F.log_softmax(self.policy_head(h), dim=1), torch.tanh(self.value_head(h))


def seeded_state_dict(seed: int) -> dict:
    """Random-init weights as `torch.manual_seed(seed); Connect4Net(2, 7)` produces them."""
    torch.manual_seed(seed)
    return Connect4Net(2, 7).state_dict()

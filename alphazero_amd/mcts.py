"""Host-side search-tree node with the reference's interface.

The engine keeps its trees on the GPU as flat 16-byte records (csrc/engine_layout.h);
this class is the object form of one such record for callers that import
`MCTSNode` (nnbattle/agents/alphazero/mcts.py:13-44).  Same attributes, same
selection score, same tie-breaking, same un-negated backup.
"""
from __future__ import annotations

import copy
import math


class MCTSNode:
    def __init__(self, parent=None, action=None, env=None):
        self.parent = parent
        self.action = action
        self.env = copy.deepcopy(env) if env else None
        self.children = {}
        self.visits = 0
        self.value = 0.0

    def is_leaf(self):
        return not self.children

    def score(self, child, c_puct):
        """value/(n+1) + c*sqrt(ln(N+1)/(n+1)) in float64 (mcts.py:29)."""
        n = child.visits + 1
        return (child.value / n) + c_puct * math.sqrt(math.log(self.visits + 1) / n)

    def best_child(self, c_puct=1.4):
        chosen, top = None, -float("inf")
        for child in self.children.values():       # insertion order = ascending column
            s = self.score(child, c_puct)
            if s > top:                            # strict: the first maximum wins
                chosen, top = child, s
        return chosen

    def expand(self, action_probs, legal_moves):
        """One child per legal move; like the reference, the probabilities are not stored."""
        for a in legal_moves:
            self.children.setdefault(a, MCTSNode(parent=self, action=a, env=self.env))

    def backpropagate(self, reward):
        node = self
        while node is not None:
            node.visits += 1
            node.value += reward
            node = node.parent


__all__ = ["MCTSNode"]

"""The binding a maintainer of the reference would add (the ctypes stub printed in INTEGRATION.md)
is executed as written, so the document cannot drift from the ABI."""
import os
import re

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_stub_runs_and_matches_the_engine():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    import numpy as np
    from alphazero_amd.engine import Engine
    from alphazero_amd.network import Connect4Net
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def make_engine" in b)
    stub = stub.replace('C.CDLL("libnnb.so")', f'C.CDLL("{os.path.join(ROOT, "alphazero_amd", "libnnb.so")}")')
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    torch.manual_seed(0)
    model = Connect4Net(2, 7)
    handle, ws = ns["make_engine"](64)
    ns["load_weights"](handle, model)
    board = np.zeros((6, 7), dtype=np.int8)
    board[5][3] = 1
    action, visits = ns["search"](handle, board, 2, 2, 64, 1.4)
    action_s, visits_s = ns["search"](handle, board, 2, 1, 64, 1.4)     # the agent still looks from colour 1
    eng = Engine(max_games=1, max_sims=64, precision="f32s")            # the stub's Config asks for float32-faithful arithmetic
    try:
        eng.load_state_dict(model.state_dict())
        root, mover = np.array([[1 << (3 * 7), 0]], dtype=np.uint64), np.array([2], dtype=np.int8)
        v, a, _, _ = eng.search(root, mover, eng.params(64, 1.4))
        vs, as_, _, _ = eng.search(root, mover, eng.params(64, 1.4), side_players=np.array([1], dtype=np.int8))
    finally:
        eng.close()
    assert action == int(a[0]) and visits == v[0].tolist() and sum(visits) == 63
    assert action_s == int(as_[0]) and visits_s == vs[0].tolist() and sum(visits_s) == 63
    assert ns["lib"].nnb_engine_destroy(handle) == 0

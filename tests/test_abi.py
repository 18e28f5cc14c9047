"""The C-ABI library loads on a machine without a GPU and exports what include/nnb.h declares.
No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nnb.h")


@pytest.fixture(scope="module")
def lib():
    from alphazero_amd import _lib
    return _lib.load()


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nnb_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from alphazero_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nnb.h but not exported by libnnb.so"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding and header disagree"


def test_struct_sizes_match_the_header(tmp_path):
    from alphazero_amd import _lib
    from alphazero_amd.engine import PLY_DTYPE, TRAJECTORY_DTYPE
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "nnb.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                   "sizeof(nnb_config),sizeof(nnb_search_params),sizeof(nnb_ply),sizeof(nnb_trajectory),"
                   "sizeof(nnb_round_stats));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(_lib.Config), C.sizeof(_lib.SearchParams), PLY_DTYPE.itemsize,
                     TRAJECTORY_DTYPE.itemsize, C.sizeof(_lib.RoundStats)]
    assert sizes[3] == 1376 == _lib.TRAJECTORY_BYTES


def test_version_and_host_only_calls(lib):
    from alphazero_amd import _lib
    assert lib.nnb_version().startswith(b"nnb ")
    cfg = _lib.Config(4096, 200, 0, 1, 1, 0)
    n = C.c_size_t()
    assert lib.nnb_workspace_bytes(C.byref(cfg), C.byref(n)) == 0
    nodes = 4096 * (1 + 7 * 200) * 16
    acts = 2 * 4096 * (56 + 72 + 90) * 128 * 4          # two leaf batches (two networks, or two halves of the slots)
    weights = 4 * (128 * 32 + 2 * 128 * 128 * 16 + 1024 * 11520 + 8 * 1024)
    assert n.value > nodes + acts + weights
    assert n.value < 1.2 * (nodes + acts + weights) + (64 << 20)


def test_argument_errors_are_reported(lib):
    from alphazero_amd import _lib
    n = C.c_size_t()
    assert lib.nnb_workspace_bytes(None, C.byref(n)) == -1
    assert b"null" in lib.nnb_last_error()
    for bad in (_lib.Config(0, 200, 0, 1, 0, 0), _lib.Config(16, 0, 0, 1, 0, 0), _lib.Config(16, 20000, 0, 1, 0, 0),
                _lib.Config(16, 200, 0, 0, 0, 0), _lib.Config(16, 200, 0, 1, 7, 0)):
        assert lib.nnb_workspace_bytes(C.byref(bad), C.byref(n)) == -1
        assert b"out of range" in lib.nnb_last_error()
    assert lib.nnb_game_ops(None, -1, None, None, None, None, None, None) == -1
    assert lib.nnb_game_ops(None, 0, None, None, None, None, None, None) == 0      # empty input is fine
    assert lib.nnb_search(None, None, None, 1, None, None, None, None, None, None, None, None) == -1


def test_engine_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from alphazero_amd.engine import Engine
    from alphazero_amd._lib import NnbError
    with pytest.raises(NnbError):
        Engine(max_games=1, max_sims=10)
    from alphazero_amd.agent import initialize_agent
    from alphazero_amd.game import ConnectFourGame
    agent = initialize_agent(load_model=False, num_simulations=10)
    with pytest.raises(NnbError):
        agent.select_move(ConnectFourGame())       # no silent CPU search


def test_product_code_never_imports_the_oracle():
    bad = []
    for base in ("alphazero_amd", "nnbattle"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    text = open(os.path.join(dirpath, f)).read()
                    if re.search(r"\boracle\b", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad

"""The two chains of a self-play round (train_alpha_zero.py:33-55 played on many slots at once) run on the caller's
stream and on a second stream of the engine.  Two HIP streams that share a hardware queue run their kernels one
after the other, and which queue a stream gets depends on the streams the process already has: with RCCL
initialised first (every rank of a multi-GPU run) the engine's second stream used to land on the caller's queue and
the two chains took turns.  The engine measures the pair before its first launch and replaces the second stream
until the pair runs side by side."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_second_stream_runs_beside_the_callers_stream():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from alphazero_amd.engine import Engine
    eng = Engine(max_games=64, max_sims=8)
    try:
        beside, replaced = eng.second_stream()
        assert beside and replaced >= 0
        other = torch.cuda.Stream()                      # the same on a stream of the caller's choosing
        with torch.cuda.stream(other):
            beside, replaced2 = eng.second_stream()
        assert beside and replaced2 >= replaced
    finally:
        eng.close()


def test_second_stream_after_a_collective_library_took_its_streams():
    """One rank over RCCL in a process of its own, as `python -m torch.distributed.run --nproc-per-node 1` sets it
    up: the pair must run side by side afterwards, and a short two-chain stream of games must equal the one-chain
    one byte for byte."""
    script = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
        from alphazero_amd.engine import Engine, trajectories_to_numpy
        from alphazero_amd.network import seeded_state_dict
        eng = Engine(max_games=64, max_sims=16)
        eng.load_state_dict(seeded_state_dict(0))
        beside, replaced = eng.second_stream()
        out = []
        for one in (False, True):
            prm = eng.params(16, 1.4, evaluator="net", mode="compat", one_stream=one, two_streams=not one)
            traj, stats = eng.selfplay_round(96, prm)
            out.append(trajectories_to_numpy(traj).tobytes())
        eng.close()
        dist.destroy_process_group()
        print("RESULT", int(beside), replaced, int(out[0] == out[1]), stats["chains"])
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=240, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-2000:])
    beside, replaced, same, _ = (int(x) for x in line[0].split()[1:])
    assert beside == 1 and same == 1

"""The engine's data-parallel branch over the real RCCL backend on the one test GPU (a one-rank "nccl" group): the
collectives the 8-GPU run uses - all_gather_into_tensor, reduce_scatter_tensor, async bucketed all_reduce with Adam
behind wait() - execute, and the step agrees with the non-distributed one.  Child process (the process group must exist
before anything touches the GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_collectives_run_in_a_one_rank_group_and_match_the_single_process_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py"), "tiny2", "16"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rccl world-1 path OK" in r.stdout

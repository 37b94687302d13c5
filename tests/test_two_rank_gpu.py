"""Two data-parallel ranks sharing the one GPU of the test box (gloo backend, CUDA tensors): the multi-rank engine path
end to end.  Runs tools/two_rank_gpu.py in a child process (it spawns the two ranks)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_identical_replicas_and_gathered_loss():
    """Embedding all-gather, key-gradient reduce-scatter, bucketed all-reduce overlapped with backward, fused clip+Adam:
    both ranks finish the step with BIT-identical parameters (the clip norm is summed in a fixed order), and the mean of
    the ranks' gathered global losses equals the single-process loss on the concatenated batch."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_rank_gpu.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "two-rank GPU path OK" in r.stdout

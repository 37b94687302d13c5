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


@pytest.mark.gpu
def test_two_ranks_with_the_globalised_local_loss_equal_one_process_on_the_concatenated_batch():
    """cfg.local_loss_global (each rank's images against the gathered captions of all ranks): the local
    loss reported by both ranks is the one-process loss on the concatenated batch, and the all-reduced gradient the optimizer sees is
    the one-process gradient (rel 2e-2) - every term of the step, not only the gathered global loss."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TWO_RANK_GLOBAL_LOCAL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_rank_gpu.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "averaged gradient against the one-process gradient" in r.stdout and "two-rank GPU path OK" in r.stdout


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_prints_the_contract_line():
    """bench.py through torch.distributed.run with two ranks on the one GPU (MEDMOE_DIST_BACKEND=gloo): rank 0 prints ONE
    JSON line with the contract's keys, n_gpus = 2, strong scaling (per-rank batch halves)."""
    import json
    env = dict(os.environ, MEDMOE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29563", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "tiny"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["per_gpu_batch"] * 2 == d["config"]["global_batch"]
    assert d["value"] > 0 and "cpu_baseline" not in d


@pytest.mark.gpu
def test_two_ranks_with_a_trainable_text_tower_keep_identical_replicas():
    """cfg.freeze_text = False under data parallelism (two gloo ranks on the one GPU): caption-side gradients of the gathered global loss come
    back through a second reduce-scatter, the text gradient is all-reduced, ONE clip norm over both towers - image AND text parameters are
    bit-identical on both ranks after the step, and the text tower moved."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TWO_RANK_TRAIN_TEXT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_rank_gpu.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "text tower under two ranks: replicas identical" in r.stdout and "two-rank GPU path OK" in r.stdout

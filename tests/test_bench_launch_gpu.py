"""GPU: ``python bench.py --gpus N`` starts its own N ranks (the driver's SCALE command has no launcher in front; the
reference spawns its workers itself as well, train.py:185-187).  On a one-GPU box the two ranks share cuda:0 and talk over
gloo (rehearsal switches of bench.py); everything else is the product launcher path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_launches_two_ranks(device):
    env = dict(os.environ, SIS_BENCH_SHARE_GPU="1", SIS_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "synthesis", "--steps", "2",
                          "--warmup", "1", "--batch", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    result = json.loads(lines[0])
    assert result["n_gpus"] == 2 and result["scaling"] == "weak" and result["value"] > 0
    assert result["config"]["batch_per_gpu"] == 4


def test_bench_gpus_2_training_workload_two_ranks(device):
    """The segmentation-training leg of ``bench.py --gpus 2``: two ranks, each with its own EMANet behind the bucketed gradient
    exchange (gloo on the shared GPU), the step eager (an exchange that is not stream work is never captured), rank 0 alone
    prints the line; nothing a single rank does on its own (CPU baseline, library-time pass, data-parallel rehearsal) may hold
    a collective -- a bench that hangs at N > 1 is worse than one that is slow."""
    env = dict(os.environ, SIS_BENCH_SHARE_GPU="1", SIS_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "emanet", "--steps", "2",
                          "--warmup", "1", "--batch", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    result = json.loads(lines[0])
    assert result["n_gpus"] == 2 and result["scaling"] == "weak" and result["value"] > 0
    assert result["library_ms_per_step"] is None and result["data_parallel_rehearsal"] is None

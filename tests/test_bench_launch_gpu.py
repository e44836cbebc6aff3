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


def test_bench_line_carries_the_contract_fields(device):
    """One short N = 1 run of the synthesis leg (CPU baseline included): ONE JSON line with every field of the bench contract --
    metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
    config.workload, the `roofline` object of the dominant kernel (bound, achieved, peak, unit, frac, traffic) and the
    `cpu_baseline` object (value, unit, cores, kind, sample)."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "synthesis", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) < 6144, f"the bench line is {len(lines[0])} bytes: per-kernel tables belong in the detail file"
    r = json.loads(lines[0])
    assert "kernels" not in r["roofline"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] >= 1 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["unit"] == "images/s" and r["dtype"] == "f32" and r["data"] == "synthetic" and r["scaling"] == "weak"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - r["config"]["batch_per_gpu"] * 1e3 / r["ms_per_step"]) <= 0.01 * r["value"]
    roof = r["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s") and "traffic" in roof
    assert roof["peak"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac"] < 1
    cpu = r["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == r["unit"] and cpu["sample"]

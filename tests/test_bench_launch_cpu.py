"""bench.py's launcher logic on the CPU: `--gpus N` without a launcher starts N ranks itself, `n_gpus` comes from
the live process group, a box with fewer GPUs than asked for is refused, and a launcher whose WORLD_SIZE disagrees
with --gpus is refused.  The workload is bench.py's `mock` (gloo, CPU tensors): only the host logic is under test."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "mock"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["parallelism"] == "x2"
    assert line["timing"]["regions"] >= 1 and line["timing"]["steps_per_region"] == 3


def test_gpus1_runs_in_process():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--workload", "mock"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1


def test_more_gpus_than_visible_is_refused():
    """the default (real) workloads: asking for more GPUs than the box has must fail loudly, not report N = 1"""
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--workload", "mock"],
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing" in r.stderr


def test_gpus8_line_schema_and_collective_preflight():
    """the N = 8 launch the driver runs at round end, rehearsed on gloo: eight ranks, ONE line with n_gpus = rccl_ranks = 8,
    and every collective primitive of the sharded paths (uneven all_to_all_single, uneven all-gather, all-reduce SUM /
    MAX) verified against known answers before the workload starts"""
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", "mock"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["config"]["parallelism"] == "x8"
    pf = line["preflight"]
    for k in ("all_reduce_sum", "all_reduce_max", "all_to_all_single_uneven", "all_gather_into_tensor", "all_gather_uneven"):
        assert pf[k] == "ok", (k, pf[k])
    assert pf["ranks_agree_all_ok"] is True
    assert r.stderr.count("rccl_ranks 8") == 8       # every rank reported its device (the ranks' lines may interleave)

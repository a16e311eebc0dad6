"""bench.py's launch paths, executed here on the CPU emulator build of the kernels (--emu, gloo):
`python bench.py --gpus 2` outside any launcher starts its own two ranks (the command the driver uses for N = 1 is the
same file, same function), and the N = 1 path carries `parity` and `cpu_baseline` from the oracle leg."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--emu", "--steps", "1", "--warmup", "0", "--windows", "1", "--height", "32", "--width", "40", "--dtype", "fp32"]


def _run(extra):
    env = dict(os.environ, MAPNET_EMU_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + TINY, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.slow
def test_bench_self_launches_two_ranks():
    d = _run(["--gpus", "2", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["n_ranks_seen"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["launcher"] == "torch.distributed.run"
    assert d["config"]["global_windows"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["unit"] == "images/s" and "cpu_baseline" not in d


@pytest.mark.slow
def test_bench_single_rank_reports_parity_and_cpu_baseline():
    d = _run(["--gpus", "1"])
    assert d["n_gpus"] == 1 and d["config"]["launcher"].startswith("none")
    p = d["parity"]
    assert p["dtype"] == "fp32" and p["loss_rel"] <= 1e-4 and p["pose_abs_max"] <= 1e-3, p
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["cpu"]


def test_bench_rejects_a_launcher_rank_count_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TINY, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "started 1 ranks" in (r.stdout + r.stderr)

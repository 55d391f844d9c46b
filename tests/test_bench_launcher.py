"""CPU: `bench.py --gpus 2` started WITHOUT torch.distributed.run must launch its own two ranks (before touching any
GPU), run one all-reduce per evaluation through `PipelinedStatsReducer(bucket=1)` over gloo and print ONE JSON line with
n_gpus = 2 (`--selftest-launcher`: CPU tensors, synthetic per-evaluation sums, no kernels).  VERDICT r1, next-round item 3."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "1",
                           "--selftest-launcher"] + extra, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_one_collective_per_evaluation():
    r = _run(["--backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["selftest"] and d["ok"]
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["scaling"] == "strong"
    assert d["rows_per_rank"] == 32768 and d["collectives_per_evaluation"] == 1


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)

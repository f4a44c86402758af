"""bench.py's N-rank launch path on the CPU: `python bench.py --gpus 2 --dry-launch` must spawn the ranks itself (no
launcher in front), let them rendezvous at 127.0.0.1 and relay ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks_over_gloo():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--workload",
                          "nfcf100m"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["sum_of_ranks"] == 3.0
    assert d["workload"] == "nfcf100m"


def test_bench_refuses_a_world_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)

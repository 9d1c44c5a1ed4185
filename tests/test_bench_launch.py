"""`python bench.py --gpus N` must start its own ranks (the driver calls it exactly like that): N child processes through
torch.distributed.run, spawned before anything touches the GPU, rank 0's JSON line forwarded, a failing rank = a failing
exit code.  Runs on the CPU box with gloo and `--dry-run` (no GPU work: the ranks only join the group and are counted)."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=600)


def test_two_ranks_without_a_launcher():
    r = _run("--gpus", "2", "--dist-backend", "gloo", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_counted"] == 2
    assert line["distributed"] == {"world_size": 2, "backend": "gloo"}
    assert "spawning" in r.stderr


def test_a_failing_rank_fails_the_launch():
    if torch.cuda.is_available():
        return  # (on a GPU box the ranks would run the benchmark: covered by the bench itself)
    r = _run("--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr

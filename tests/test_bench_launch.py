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


def test_a_rank_that_leaves_with_an_error_fails_the_launch_with_or_without_a_gpu():
    """The exit-code path of the self-launch on any box: in --dry-run nothing touches a GPU, rank 1 leaves with an error after
    the ranks were counted, and `python bench.py --gpus 2` must not return 0."""
    r = _run("--gpus", "2", "--dist-backend", "gloo", "--dry-run", "--dry-run-fail-rank", "1")
    assert r.returncode != 0, r.stdout[-500:]


def test_eight_ranks_and_one_gather_round_without_a_launcher():
    """The launch the driver makes on an 8-GPU node, rehearsed with gloo on the CPU box: eight ranks counted, one round of
    config 4's observation gather in both modes with unequal controlled counts, every rank's rows in its own section --
    so that the first RCCL run of the class is not also its first 8-rank run."""
    r = _run("--gpus", "8", "--dist-backend", "gloo", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 8 and line["ranks_counted"] == 8
    assert line["gather_round"]["ok"] is True and len(line["gather_round"]["controlled_per_rank"]) == 8
    t = line["gather_bytes_per_rank"]
    assert t["raw_rows"]["bytes_per_agent"] < t["packed"]["bytes_per_agent"]
    assert t["packed"]["raw"] == 1024 * 64 * 4 * (6 + 63 * 6 + 200 * 13)


def _processes():
    """(pid, argv) of every live process, from /proc."""
    out = []
    for pid in os.listdir("/proc"):
        if pid.isdigit():
            try:
                with open("/proc/%s/cmdline" % pid, "rb") as fh:
                    argv = fh.read().split(b"\0")
                with open("/proc/%s/stat" % pid) as fh:
                    if fh.read().rsplit(")", 1)[1].split()[0] == "Z":
                        continue
                out.append((pid, [a.decode("utf-8", "replace") for a in argv]))
            except OSError:
                pass
    return out


def test_a_signal_to_the_launcher_ends_every_rank():
    """`timeout` around bench.py (tools, the driver) signals only the parent: the ranks live in a session of their own and must
    be ended with it, not left on the GPUs."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--dry-run",
                          "--dry-run-sleep", "120"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 120
    kids = []
    while time.time() < deadline and len(kids) < 3:   # the launcher (torch.distributed.run) and its two ranks (which it starts in
        time.sleep(1.0)                               # sessions of their own and ends when it is told to end)
        kids = [pid for pid, cmd in _processes()
                if "--dry-run-sleep" in cmd and pid != str(p.pid) and ("torch.distributed.run" in cmd or "-u" in cmd)]
    assert len(kids) >= 3, "the ranks did not start: %r %r %s" % (kids, p.poll(), (p.stderr.read()[-800:] if p.poll() is not None else ""))
    time.sleep(3.0)   # (the ranks have joined the group and sleep)
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    for _ in range(20):
        alive = [pid for pid, _ in _processes()]
        left = [k for k in kids if k in alive]
        if not left:
            break
        time.sleep(1.0)
    assert not left, "ranks survived the launcher"

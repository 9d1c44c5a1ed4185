"""CPU suite: the N>1 path (world sharding + optional observation all-gather) on a world_size-2
gloo process group.  No GPU, no simulator: the sharding layer is pure torch.distributed."""
import os
import socket
import subprocess
import sys
import textwrap

from gpudrive_lab_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_all_worlds():
    for total in (8192, 1024, 10, 7):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a < b <= d
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_scene_tiling_continues_across_ranks():
    scenes = ["a", "b", "c"]
    r0 = sharding.scene_list_for_rank(scenes, 4, 0)
    r1 = sharding.scene_list_for_rank(scenes, 4, 1)
    assert r0 == ["a", "b", "c", "a"] and r1 == ["b", "c", "a", "b"]


_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    from gpudrive_lab_amd import sharding
    rank, local_rank, world = sharding.init_process_group(backend="gloo")
    assert world == 2 and torch.distributed.is_initialized()
    dev = torch.device("cpu")
    # each rank owns 3 worlds; obs block [W_local, A, F] filled with its global world index
    lo, hi = sharding.shard_range(6, rank, world)
    assert (lo, hi) == ((0, 3) if rank == 0 else (3, 6))
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 5).contiguous()
    sharding.barrier(dev)
    full = sharding.gather_observations(local)
    assert full.shape == (6, 4, 5)
    assert torch.equal(full[:, 0, 0], torch.arange(6, dtype=torch.float32))
    # bench aggregation: MAX of the elapsed time, SUM of the live agents
    assert sharding.reduce_max(1.0 + rank, dev) == 2.0
    assert sharding.reduce_sum(10 + rank, dev) == 21.0
    # preallocated output buffer is reused
    out = torch.empty(6, 4, 5)
    assert sharding.gather_observations(local, out).data_ptr() == out.data_ptr()
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_two_rank_gloo_gather_and_reductions(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=180)
        outs.append(out)
        assert p.returncode == 0, out
    assert all("ok" in o for o in outs)


def test_scene_tiling_matches_reference_dataloader_golden():
    """Rank 0's scene list is what the reference's SceneDataLoader yields when the dataset is smaller than the
    batch (tests/golden/make_dataset_tiling_golden.py, generated with gpudrive/env/dataset.py itself); the
    other ranks continue the same round-robin where the previous rank stopped."""
    import json
    import os
    from gpudrive_lab_amd import sharding
    root = os.path.dirname(os.path.abspath(__file__))
    for case in json.load(open(os.path.join(root, "golden", "dataset_tiling_golden.json"))):
        files = list(range(case["n_files"]))
        assert sharding.scene_list_for_rank(files, case["batch"], 0) == case["indices"]
        two = sharding.scene_list_for_rank(files, case["batch"], 0) + sharding.scene_list_for_rank(files, case["batch"], 1)
        assert two == [files[i % len(files)] for i in range(2 * case["batch"])]

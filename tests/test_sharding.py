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
    # config 4: the packed observation [W_local, A, D] of the real shape (A = 64: D = 6 + 63 * 6 + 200 * 13), raw and
    # controlled-agent-compacted, double-buffered; rank r's rows must land in section r, in agent order
    W, A, D = 3, 64, 6 + 63 * 6 + 200 * 13
    g = torch.Generator().manual_seed(100 + rank)
    obs = torch.rand(W, A, D, generator=g)
    ctrl = torch.rand(W, A, generator=g) < (0.3 if rank == 0 else 0.6)   # unequal counts across ranks
    peer = torch.Generator().manual_seed(100 + (1 - rank))
    peer_obs = torch.rand(W, A, D, generator=peer)
    peer_ctrl = torch.rand(W, A, generator=peer) < (0.3 if rank == 1 else 0.6)
    both = {rank: (obs, ctrl), 1 - rank: (peer_obs, peer_ctrl)}
    raw = sharding.ObservationGather("raw", W * A, D, dev)
    raw.start(obs)
    full, counts = raw.wait()
    assert full.shape == (2 * W * A, D) and counts.tolist() == [W * A, W * A] and raw.bytes_per_rank == W * A * D * 4
    for r in range(2):
        assert torch.equal(full[r * W * A:(r + 1) * W * A], both[r][0].reshape(-1, D))
    cg = sharding.ObservationGather("compact", W * A, D, dev)
    cg.set_mask(ctrl)
    n = [int(both[r][1].sum()) for r in range(2)]
    assert cg.counts.tolist() == n and cg.cap == max(n)
    for step in range(3):   # the two buffers alternate
        cg.start(obs + step)
        full, counts = cg.wait()
        assert full.shape == (2 * cg.cap, D)
        for r in range(2):
            exp = (both[r][0] + step).reshape(-1, D)[both[r][1].reshape(-1)]
            assert torch.equal(full[r * cg.cap:r * cg.cap + n[r]], exp), (step, r)
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

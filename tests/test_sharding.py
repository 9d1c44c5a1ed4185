"""CPU suite: the N>1 path (world sharding + optional observation all-gather) on a world_size-2
gloo process group.  No GPU, no simulator: the sharding layer is pure torch.distributed."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

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
    backend = os.environ.get("GD_TEST_BACKEND", "gloo")
    rank, local_rank, world = sharding.init_process_group(backend=backend)
    assert world == 2 and torch.distributed.is_initialized() and torch.distributed.get_backend() == backend
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    # each rank owns 3 worlds; obs block [W_local, A, F] filled with its global world index
    lo, hi = sharding.shard_range(6, rank, world)
    assert (lo, hi) == ((0, 3) if rank == 0 else (3, 6))
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 5).contiguous().to(dev)
    sharding.barrier(dev)
    full = sharding.gather_observations(local)
    assert full.shape == (6, 4, 5)
    assert torch.equal(full[:, 0, 0].cpu(), torch.arange(6, dtype=torch.float32))
    # bench aggregation: MAX of the elapsed time, SUM of the live agents
    assert sharding.reduce_max(1.0 + rank, dev) == 2.0
    assert sharding.reduce_sum(10 + rank, dev) == 21.0
    # preallocated output buffer is reused
    out = torch.empty(6, 4, 5, device=dev)
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
    obs_d, ctrl_d = obs.to(dev), ctrl.to(dev)
    raw = sharding.ObservationGather("raw", W * A, D, dev)
    raw.start(obs_d)
    full, counts = raw.wait()
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    assert full.shape == (2 * W * A, D) and counts.tolist() == [W * A, W * A] and raw.bytes_per_rank == W * A * D * 4
    for r in range(2):
        assert torch.equal(full[r * W * A:(r + 1) * W * A].cpu(), both[r][0].reshape(-1, D))
    cg = sharding.ObservationGather("compact", W * A, D, dev, timing=True, timing_window=4)
    cg.set_mask(ctrl_d)
    n = [int(both[r][1].sum()) for r in range(2)]
    assert cg.counts.tolist() == n and cg.cap == max(n)

    def check(full, offset):
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        assert full.shape == (2 * cg.cap, D)
        for r in range(2):
            exp = (both[r][0] + offset).reshape(-1, D)[both[r][1].reshape(-1)]
            assert torch.equal(full[r * cg.cap:r * cg.cap + n[r]].cpu(), exp), (offset, r)
    for step in range(3):   # the two buffers alternate
        cg.start(obs_d + step)
        full, counts = cg.wait()
        check(full, step)
    # start() owns the buffer invariant: starts without a wait() in between (each re-uses the pair of two starts
    # back), then the last one's result; the timing window is bounded
    for step in range(6):
        cg.start(obs_d + 10 + step)
    full, counts = cg.wait()
    check(full, 15)
    assert len(cg.events) <= 4
    # a new mask right after a start(): the in-flight gather is drained before the buffers are replaced
    cg.start(obs_d + 20)
    cg.set_mask(ctrl_d)
    cg.start(obs_d + 21)
    full, counts = cg.wait()
    check(full, 21)
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""")


def _run_two_ranks(tmp_path, backend):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", GD_TEST_BACKEND=backend,
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=180)
        outs.append(out)
        assert p.returncode == 0, out
    assert all("ok" in o for o in outs)


def test_two_rank_gloo_gather_and_reductions(tmp_path):
    _run_two_ranks(tmp_path, "gloo")


@pytest.mark.gpu
def test_two_rank_rccl_gather_matches_the_gloo_expectations(tmp_path):
    """BASELINE configs[3]: the same worker on a 2-rank nccl (= RCCL) group, one GPU per rank -- the first time RCCL sees
    ObservationGather is under a test, on whatever box has two GPUs (the builder's box has one: skipped there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_two_ranks(tmp_path, "nccl")


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


_WORKER8 = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    from gpudrive_lab_amd import sharding
    rank, local_rank, world = sharding.init_process_group(backend="gloo")
    assert world == 8
    dev = torch.device("cpu")
    W, A, D = 2, 64, 6 + 63 * 6 + 200 * 13
    lo, hi = sharding.shard_range(8 * W, rank, world)
    assert (lo, hi) == (rank * W, rank * W + W)
    obs = torch.full((W, A, D), float(rank)) + torch.arange(W * A, dtype=torch.float32).view(W, A, 1) / 1024.0
    g = torch.Generator().manual_seed(40 + rank)
    ctrl = torch.rand(W, A, generator=g) < 0.1 + 0.1 * rank          # 10 to 80 percent controlled: very unequal blocks
    for mode in ("raw", "compact"):
        og = sharding.ObservationGather(mode, W * A, D, dev)
        og.set_mask(ctrl)
        for step in range(3):
            og.start(obs + step)
            full, counts = og.wait()
        assert full.shape == (8 * og.cap, D) and len(counts) == 8
        for r in range(8):
            gr = torch.Generator().manual_seed(40 + r)
            cr = torch.rand(W, A, generator=gr) < 0.1 + 0.1 * r
            exp = (torch.full((W, A, D), float(r)) + torch.arange(W * A, dtype=torch.float32).view(W, A, 1) / 1024.0 + 2).reshape(-1, D)
            if mode == "compact":
                exp = exp[cr.reshape(-1)]
                assert int(counts[r]) == int(cr.sum())
            assert torch.equal(full[r * og.cap:r * og.cap + exp.shape[0]], exp), (mode, r)
    assert sharding.reduce_sum(1, dev) == 8.0 and sharding.reduce_max(rank, dev) == 7.0
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_eight_rank_gloo_gather(tmp_path):
    """Config 4's shape of job -- 8 ranks, contiguous world blocks, the observation gathered in both modes with controlled
    shares from 10 % to 80 % -- on gloo, so that the first RCCL run of ObservationGather is not also its first 8-rank run."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8 % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and "ok" in out, out

"""World sharding across the GPUs of one node (one process per GPU).

Worlds are fully independent (no cross-world reads anywhere in reference src/sim.cpp), so rank r
owns a contiguous block of worlds and the step needs NO per-step exchange.  The only collective is
optional: an all-gather of the observation block for a single-process learner (RCCL over xGMI when
the backend is "nccl"; gloo in the CPU tests).  Every peer's shard travels over its own direct xGMI
link in an all-gather of equal blocks, so the time is about shard_bytes / 153 GB/s, not a ring's
7 * shard_bytes / 153 GB/s.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_range(total_worlds, rank, world):
    """Contiguous block of worlds owned by `rank`: [lo, hi).  Remainders go to the low ranks."""
    base, rem = divmod(total_worlds, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def scene_list_for_rank(all_scenes, worlds_per_rank, rank):
    """Round-robin tiling of the scene list over a rank's worlds, continuing where the previous rank
    stopped (what SceneDataLoader does when dataset < batch, gpudrive/env/dataset.py:64-67)."""
    n = len(all_scenes)
    start = rank * worlds_per_rank
    return [all_scenes[(start + i) % n] for i in range(worlds_per_rank)]


def _collective_device(device):
    """RCCL reduces device tensors; gloo (CPU tests, single-GPU rehearsals) reduces host tensors."""
    if dist.is_initialized() and dist.get_backend() == "nccl" and device is not None and device.type == "cuda":
        return device
    return torch.device("cpu")


def barrier(device=None):
    if dist.is_initialized():
        if dist.get_backend() == "nccl" and device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def reduce_max(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_observations(local, out=None):
    """All-gather a rank's observation block [W_local, ...] into [world * W_local, ...] on every
    rank (equal W_local per rank).  `out` may be a preallocated buffer reused across steps."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out

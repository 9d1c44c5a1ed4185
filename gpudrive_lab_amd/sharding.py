"""World sharding across the GPUs of one node (one process per GPU).

Worlds are fully independent (no cross-world reads anywhere in reference src/sim.cpp), so rank r
owns a contiguous block of worlds and the step needs NO per-step exchange.  The only collective is
optional: an all-gather of the observation block for a single-process learner (RCCL over xGMI when
the backend is "nccl"; gloo in the CPU tests).  Every peer's shard travels over its own direct xGMI
link in an all-gather of equal blocks, so the time is about shard_bytes / 153 GB/s, not a ring's
7 * shard_bytes / 153 GB/s.
"""
import collections
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


class ObservationGather:
    """Config 4 (BASELINE.json): the learner-side observation of every rank's worlds on every rank, gathered while
    the next step runs.

    `mode` "raw": the rank's whole block [W_local, A, D].  "compact": only the rows of controlled agents (the rows
    a learner consumes, gpudrive/env/env_puffer.py:169-176), packed to the front of a [cap, D] buffer; `cap` is the
    largest controlled-agent count over the ranks (one all-reduce at set-up), so every rank contributes an equal block
    and each peer's shard rides its own xGMI link (direct gather, not a ring).  The controlled mask is
    fixed between world rebuilds, so the row index is computed once (`set_mask`).

    `start(block)` copies / compacts `block` into a send buffer on the caller's stream and launches the all-gather on
    a side stream; `wait()` makes the caller's stream wait for it and returns (gathered [world * cap, D], counts
    [world]) -- rows beyond counts[r] in rank r's section are stale padding.  Two send / receive buffers alternate, so
    the gather of step k overlaps step k + 1; `start` itself makes the caller's stream wait for the gather that last
    used the buffer pair it is about to overwrite (two starts back), whether or not the caller waited for it.  The
    receive buffer `wait()` returned stays valid until the second `start` after it.  `timing=True` keeps the (start,
    stop) events of the last `timing_window` gathers for `mean_ms()`.  On CPU tensors (gloo, tests) everything runs inline."""

    def __init__(self, mode, world_agents, feature_dim, device, dtype=torch.float32, timing=False, timing_window=256):
        assert mode in ("raw", "compact")
        self.mode, self.device, self.dim = mode, torch.device(device), int(feature_dim)
        self.rows = int(world_agents)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.cuda = self.device.type == "cuda"
        self.nccl = dist.is_initialized() and dist.get_backend() == "nccl" and self.cuda
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.index, self.count, self.cap = None, self.rows, self.rows
        self.counts = torch.full((self.world,), self.rows, dtype=torch.int64)
        self._dtype = dtype
        self._alloc()
        self._k, self._pending = 0, None
        self._done = [None, None]  # completion event of the last gather that used buffer pair k
        self.timing = bool(timing)
        self.events = collections.deque(maxlen=int(timing_window))  # (start, stop) cuda events, only with timing=True

    def _alloc(self):
        mk = lambda n: torch.empty((n, self.dim), dtype=self._dtype, device=self.device)
        self.send = [mk(self.cap), mk(self.cap)]
        self.recv = [mk(self.world * self.cap), mk(self.world * self.cap)]

    def set_mask(self, controlled):
        """controlled: bool / int tensor [W_local, A] (controlled_state_tensor).  Compact mode only."""
        if self.mode != "compact":
            return
        flat = controlled.reshape(-1).to(torch.bool)
        self.index = torch.nonzero(flat, as_tuple=False).reshape(-1).to(self.device)
        self.count = int(self.index.numel())
        c = torch.tensor([self.count], dtype=torch.int64, device=_collective_device(self.device if self.nccl else None))
        if dist.is_initialized():
            lst = [torch.zeros_like(c) for _ in range(self.world)]
            dist.all_gather(lst, c)
            self.counts = torch.cat([t.cpu() for t in lst])
        else:
            self.counts = c.cpu()
        self.cap = max(int(self.counts.max().item()), 1)
        # a gather of the previous start() may still be running on the side stream with the old buffers
        if self.stream is not None:
            self.stream.synchronize()
        self._pending, self._done = None, [None, None]
        self._alloc()

    @property
    def bytes_per_rank(self):
        return self.cap * self.dim * self.send[0].element_size()

    def start(self, block):
        k = self._k
        self._k ^= 1
        if self._done[k] is not None:  # the gather two starts back may still be reading send[k] / writing recv[k]
            torch.cuda.current_stream(self.device).wait_event(self._done[k])
        src = block.reshape(-1, self.dim)
        if self.mode == "compact":
            torch.index_select(src, 0, self.index, out=self.send[k][:self.count])
        else:
            self.send[k].copy_(src)
        if not dist.is_initialized():
            self._pending = (k, None)
            self.recv[k][:self.cap].copy_(self.send[k])
            return
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                ev0, ev1 = torch.cuda.Event(enable_timing=self.timing), torch.cuda.Event(enable_timing=self.timing)
                if self.timing:
                    ev0.record(self.stream)
                if self.nccl:
                    dist.all_gather_into_tensor(self.recv[k], self.send[k])
                else:  # gloo rehearsal on one GPU: through host memory
                    host = torch.empty((self.world * self.cap, self.dim), dtype=self._dtype)
                    dist.all_gather_into_tensor(host, self.send[k].cpu())
                    self.recv[k].copy_(host)
                ev1.record(self.stream)
                if self.timing:
                    self.events.append((ev0, ev1))
            self._pending = (k, ev1)
            self._done[k] = ev1
        else:
            dist.all_gather_into_tensor(self.recv[k], self.send[k])
            self._pending = (k, None)

    def wait(self):
        k, ev = self._pending
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return self.recv[k], self.counts

    def mean_ms(self):
        if not self.events:
            return None
        self.events[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)


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

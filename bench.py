#!/usr/bin/env python3
"""Benchmark driver of the MI355X-native GPUDrive step engine (the `headless` counterpart,
reference src/headless.cpp:125-155).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (movement -> collision -> reward/done -> self / partner /
k-NN road observations) over every world of the rank, with seeded random actions already resident
in HBM, and a reset of all worlds every 91 steps (episode length).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1]): 1024 worlds x 64 agents per GPU, classic bicycle dynamics,
63-partner + 200-road-point k-NN observation, radius 50, collisions ignored.  Primary: the seeded
synthetic exact-64 scenes of SURVEY.md section 8d (64 live agents, 4096 road segments per world).
Secondary (reported under "other_workloads"): the committed Waymo scenes tiled round-robin.

Two stretches of K steps per workload, back to back on the same simulator:
  1. the TIMED REGION (`ms_per_step`, `value`): `step()` exactly as a user calls it -- the step's kernels replayed from
     the captured hipGraph, no instrumentation;
  2. the same K steps again with HIP events on the engine's stream around every launch (`gd_kernel_timing_*`: the
     kernels are then launched one by one): `kernels`, `roofline.avg_kernel_us`, and that stretch's own wall clock
     `ms_per_step_events`, which the kernel averages add up to.
Each stretch starts at the first step of an episode (after untimed spin-up steps, `spin_up_steps` in the line), so that it
holds K // 91 episode resets in every run (`timed_region.episode_resets`).
With N > 1 ranks a further stretch measures BASELINE configs[3]'s observation all-gather (RCCL), overlapped
with the following step (`allgather` in the line; `--gather none` skips it).  `--headless` adds the two FPS lines of the
reference's own benchmark CLI (src/headless.cpp:145-155) on stderr.
"""
import argparse
import gc
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpudrive_lab_amd import sharding, synth  # noqa: E402

# Python's cyclic collector: a full (generation 2) collection walks every container object of the process -- several
# hundred thousand once torch is imported -- and stops the host for tens of milliseconds at a moment of its own choosing.
# Inside a 20-step timed region that is longer than the region itself, so the collector is run before each timed
# stretch and switched off during it (what `timeit` does); every collection that still starts inside one is reported
# (`gc_ms_in_timed_stretches`).  GPUDRIVE_BENCH_KEEP_GC=1 leaves the collector alone.
_GC_LOG = []
_GC_T0 = [0.0]


def _gc_callback(phase, info):
    if phase == "start":
        _GC_T0[0] = time.perf_counter()
    else:
        _GC_LOG.append((_GC_T0[0], time.perf_counter(), info.get("generation", -1)))


gc.callbacks.append(_gc_callback)
KEEP_GC = os.environ.get("GPUDRIVE_BENCH_KEEP_GC") == "1"

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
EPISODE = 91



def source_stamp():
    """sha256 over the engine sources: profiles/*_traffic.json carries the stamp of the build its counters were
    collected on, and `roofline.traffic` is null when it is not this build's."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gpudrive_lab_amd", "csrc")
    for name in sorted(os.listdir(d)) + ["../../include/gpudrive_amd.h"]:
        if name.endswith((".hip", ".cpp", ".hpp", ".h")) or name == "Makefile":
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


WAYMO = [os.path.join(ROOT, "tests", "data", n) for n in
         ("test.json", "tfrecord-00002-of-01000_407.json", "tfrecord-00000-of-01000_4.json")]


def split_workload(name):
    """'synthetic_set' -> ('synthetic', 1, None): the same scenes with gd_config.knn_order = GD_KNN_SET_ORDER;
    'synthetic_128' -> ('synthetic', 0, 128): the same generator with the fork's own kMaxAgentCount = 128 agent slots
    (reference src/consts.hpp:11), every one of them live.  A '_linear' suffix stays part of the workload's name
    ('synthetic_linear', 'waymo_linear': the same scenes with roadObservationAlgorithm = AllEntitiesWithRadiusFiltering,
    the `EnvConfig` default, reference gpudrive/env/config.py:51); 'ppo_default' is what the reference's PPO baselines run
    (params_for) and always has the fork's 128 agent slots."""
    order, agents = 0, None
    if name.endswith("_set"):
        name, order = name[:-4], 1
    if name.endswith("_128"):
        name, agents = name[:-4], 128
    if name == "ppo_default":
        agents = 128
    return name, order, agents


def scene_family(workload):
    """'synthetic_linear' -> 'synthetic': which scenes a workload runs on."""
    return workload[:-7] if workload.endswith("_linear") else workload


def params_for(workload):
    kw = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0,
              dynamicsModel=0, roadObservationAlgorithm=0, isStaticAgentControlled=1,
              initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
    # "waymo_raw": the Waymo tiles with unreduced polylines (5-10 thousand roads per world; not in the default list)
    kw["polylineReductionThreshold"] = 0.0 if scene_family(workload) in ("synthetic", "rl_loop", "waymo_raw") else 0.1
    if workload.endswith("_linear") or workload == "ppo_default":
        kw["roadObservationAlgorithm"] = 1  # AllEntitiesWithRadiusFiltering (reference src/sim.cpp:258-279)
    if workload == "lidar":  # BASELINE configs[4]: LiDAR 3 x 50 rays, mixed vehicle / cyclist / pedestrian agents
        kw["enableLidar"] = 1
    if workload == "cfg3":   # BASELINE configs[2]: 4 x the worlds, collisions stop agents, goal-reach reward
        kw["collisionBehaviour"] = 0
    if workload == "ppo_default":
        # what `baselines/ppo/ppo_pufferlib.py` with `baselines/ppo/config/ppo_base_puffer.yaml` constructs through an unchanged
        # gpudrive/env/base_env.py:96-159: EnvConfig.road_obs_algorithm = "linear" (config.py:51), init_mode = "all_non_trivial"
        # (config.py:137-139: parked cars stay Static, only agents valid at t = 0 exist), remove_non_vehicles = True,
        # polyline_reduction_threshold 0.1, collision_behavior "ignore", dynamics "classic", kMaxAgentCount = 128
        kw.update(isStaticAgentControlled=0, initOnlyValidAgentsAtFirstStep=1, IgnoreNonVehicles=1)
    return kw


def scenes_for(workload, worlds, rank, agents=64):
    if scene_family(workload) in ("synthetic", "rl_loop"):
        d = os.path.join(tempfile.gettempdir(), "gpudrive_amd_bench_scenes" + ("" if agents == 64 else "_%d" % agents))
        paths = synth.write_scenes(d, [rank * 1000 + i for i in range(8)], n_agents=agents)
        return [paths[i % len(paths)] for i in range(worlds)]
    return sharding.scene_list_for_rank(WAYMO, worlds, rank)


def make_sim(scenes, kw, agents, device_index, knn_order=0, lidar_half_angle=0.0, enable_bev=False):
    import madrona_gpudrive as mg
    p = mg.Parameters()
    for k, v in kw.items():
        if k in ("rewardType", "distanceToGoalThreshold", "distanceToExpertThreshold"):
            setattr(p.rewardParams, k, v)
        else:
            setattr(p, k, v)
    return mg.SimManager(exec_mode=mg.madrona.ExecMode.CUDA, gpu_id=device_index, scenes=scenes, params=p,
                         max_agents=agents, knn_order=knn_order, lidar_half_angle=lidar_half_angle, enable_bev=enable_bev)


def action_batches(worlds, agents, device, seed, n=8):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for _ in range(n):
        a = torch.zeros(worlds, agents, 10)
        a[..., 0] = torch.rand(worlds, agents, generator=g) * 5.0 - 3.0   # U(-3, 2), src/headless.cpp:69
        a[..., 1] = torch.rand(worlds, agents, generator=g) * 1.4 - 0.7   # U(-0.7, 0.7), :70
        out.append(a.to(device))
    return out


def run_steps(sim, batches, all_worlds, n, start=0, tracker=None):
    act = sim.action_tensor().to_torch()
    for k in range(start, start + n):
        act.copy_(batches[k % len(batches)])
        if tracker is not None:
            # "rl_loop": the step as a learner sees it: simulator step, episode
            # bookkeeping + device-driven reset of finished worlds, fused observation pack; no host sync
            tracker.step()
            sim.packed_observations()
            continue
        sim.step()
        if (k + 1) % EPISODE == 0:
            sim.reset(all_worlds)
    return start + n


def linear_roads_scanned(sim, radius, k=200, chunk=16):
    """Linear mode's algorithmic scan length, from the exported tensors of the state the run ended in (plain distances: a road
    exactly at the radius may count either way): for every live agent the number of roads the reference's loop visits
    (reference src/sim.cpp:261-275: up to and including the K-th road within the radius, or every road of the world).
    Returns (sum over the live agents, sum over the worlds of the longest such prefix among the world's agents)."""
    shape = sim.shape_tensor().to_torch()
    pos = sim.absolute_self_observation_tensor().to_torch()[..., 0:2]
    roads = sim.map_observation_tensor().to_torch()[..., 0:2]
    W, A = pos.shape[0], pos.shape[1]
    total, per_world = 0, 0
    for w0 in range(0, W, chunk):
        w1 = min(W, w0 + chunk)
        n = shape[w0:w1, 0]
        R = shape[w0:w1, 1]
        rmax = int(R.max().item())
        if rmax == 0:
            continue
        d2 = ((pos[w0:w1, :, None, :] - roads[w0:w1, None, :rmax, :]) ** 2).sum(-1)           # [w, A, rmax]
        valid = torch.arange(rmax, device=d2.device)[None, None, :] < R[:, None, None]
        reached = ((d2 <= radius * radius) & valid).cumsum(-1) >= k
        first = torch.where(reached.any(-1), reached.int().argmax(-1) + 1, R[:, None].expand(-1, A).int())
        live = torch.arange(A, device=d2.device)[None, :] < n[:, None]
        total += int((first * live).sum().item())
        per_world += int((first * live).max(dim=1).values.sum().item())
    return total, per_world


def bench_workload(workload, args, rank, local_rank, world, device):
    # everything (action writes, steps, resets) runs on one side stream: the engine replays its step as
    # a hipGraph there (the legacy null stream cannot be captured)
    with torch.cuda.stream(torch.cuda.Stream(device=device)):
        return _bench_workload(workload, args, rank, local_rank, world, device)


def _bench_workload(name, args, rank, local_rank, world, device):
    workload, knn_order, agents_override = split_workload(name)
    knn_order = max(knn_order, args.knn_order)
    kw = params_for(workload)
    linear = kw["roadObservationAlgorithm"] == 1
    if workload == "cfg3":
        args = argparse.Namespace(**dict(vars(args), worlds=4 * args.worlds))
    if agents_override:
        args = argparse.Namespace(**dict(vars(args), agents=agents_override))
    scenes = scenes_for(workload, args.worlds, rank, args.agents)
    t0 = time.time()
    sim = make_sim(scenes, kw, args.agents, local_rank, knn_order=knn_order,
                   lidar_half_angle=float(np.pi) if workload == "lidar" else 0.0,  # 360 degrees
                   enable_bev=workload == "bev")  # the reference rasterises the 200 x 200 BEV on every step
    torch.cuda.synchronize(device)
    init_s = time.time() - t0
    shape = sim.shape_tensor().to_torch().cpu().numpy()
    live = int(shape[:, 0].sum())
    roads = int(shape[:, 1].sum())
    batches = action_batches(args.worlds, args.agents, device, seed=1234 + rank)
    all_worlds = np.arange(args.worlds, dtype=np.int32)
    tracker, direct = None, False
    if workload == "rl_loop":
        from gpudrive_lab_amd.episode import EpisodeTracker
        tracker = EpisodeTracker(sim)
        # the packed observation is written where the rows are produced, and the raw partner / road rows -- which nothing in this
        # loop reads -- are not written at all (gd_attach_packed); GPUDRIVE_BENCH_SECOND_PASS=1 keeps rounds 2-4's second pass
        # over the raw tensors (k_pack_obs)
        direct = os.environ.get("GPUDRIVE_BENCH_SECOND_PASS") != "1" and sim.direct_pack(only=True)

    gc_ms = [0.0]
    spin_steps = [0]

    def timed_stretch(k, pre_t0=None):
        # Device spin-up: untimed steps for --spin-ms of wall clock right before the stretch.  After an idle period (the host
        # builds the worlds for a second or more) this GPU delays the first work it is given by 10-90 ms in about one
        # stretch out of twelve -- the kernels then run at full speed, the wall clock of a 20-step stretch triples; with
        # the device kept busy for 200 ms first it did not happen in 60 stretches (tools/stall_probe2.py).
        if not KEEP_GC:  # (collected before the spin-up, not after it: the device must not idle between the spin-up and t0)
            gc.collect()
            gc.disable()
        t_spin = time.perf_counter()
        while 1e3 * (time.perf_counter() - t_spin) < args.spin_ms:
            k = run_steps(sim, batches, all_worlds, 4, start=k, tracker=tracker)
            spin_steps[0] += 4
            torch.cuda.synchronize(device)
        # ... and on to the start of an episode, so that the stretch holds the same number of episode resets (every 91st step:
        # a second observation pass, 1.4 ms on the bench scene) in every run: K // 91 of them.  With a time-based spin-up
        # alone the driver's 20-step stretch caught one in some runs (1.30 ms per step) and none in others (1.25)
        # (--no-align: profiler passes, whose counters would otherwise average over up to 90 alignment steps per stretch.
        # With N ranks every rank steps on to the furthest rank's k, so that all of them reach the barrier together.)
        if not args.no_align:
            target = int(sharding.reduce_max(float(-(-k // EPISODE) * EPISODE), device)) if world > 1 else -(-k // EPISODE) * EPISODE
            while k < target:
                k = run_steps(sim, batches, all_worlds, 1, start=k, tracker=tracker)
                spin_steps[0] += 1
        torch.cuda.synchronize(device)
        if pre_t0 is not None:
            pre_t0()  # counters that must cover the K timed steps and nothing else start here
        sharding.barrier(device)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        k = run_steps(sim, batches, all_worlds, args.steps, start=k, tracker=tracker)
        torch.cuda.synchronize(device)
        sharding.barrier(device)
        t1 = time.perf_counter()
        gc.enable()
        gc_ms[0] += 1e3 * sum(min(b, t1) - max(a, t0) for a, b, _ in _GC_LOG if b > t0 and a < t1)
        return k, t1 - t0

    k = run_steps(sim, batches, all_worlds, args.warmup, tracker=tracker)
    # ---- 1. the timed region: step() as a user calls it (hipGraph replay, no instrumentation) ----
    base = {}
    k, local_elapsed = timed_stretch(k, pre_t0=lambda: base.update(graph=sim.stat(0), plain=sim.stat(1)))
    graph_steps, plain_steps = sim.stat(0) - base["graph"], sim.stat(1) - base["plain"]
    elapsed = sharding.reduce_max(local_elapsed, device)
    elapsed_min = -sharding.reduce_max(-local_elapsed, device)
    # ---- 2. the same K steps again with HIP events around every kernel launch (kernel by kernel); a few untimed steps
    # first so that nothing of the switch (event creation, first plain launches) lands in the stretch ----
    sim.kernel_timing(True)
    k = run_steps(sim, batches, all_worlds, 3, start=k, tracker=tracker)
    # the sums are zeroed after the spin-up and the alignment steps, right before the barrier and t0: `launches` == K
    # (+ the stretch's episode resets) and `kernels_sum_us` is comparable with `ms_per_step_events`
    k, ev_elapsed = timed_stretch(k, pre_t0=lambda: (sim.kernel_timing(True), sim.stat(30)))  # (reading the skip counter zeroes it)
    ev_elapsed = sharding.reduce_max(ev_elapsed, device)
    rows_skipped = sim.stat(30)  # agents whose road rows were left in place over the events stretch (pose bits unchanged)
    total_live = sharding.reduce_sum(live, device)
    res = dict(
        workload=name, seconds=elapsed, ms_per_step=1e3 * elapsed / args.steps,
        ms_per_step_events=1e3 * ev_elapsed / args.steps,
        ms_per_step_min_rank=1e3 * elapsed_min / args.steps,
        knn_order=knn_order,
        live_agents_per_rank=live, road_entities_per_rank=roads, init_seconds=init_s,
        agent_steps_per_s=total_live * args.steps / elapsed,
        padded_agent_steps_per_s=world * args.worlds * args.agents * args.steps / elapsed,
        timed_region=dict(graph_steps=graph_steps, plain_steps=plain_steps, episode_resets=args.steps // EPISODE),
        gc_ms_in_timed_stretches=gc_ms[0], spin_up_steps=spin_steps[0],
        worlds=args.worlds, packed_observation="written by the step (gd_attach_packed, raw partner / road rows not written)" if direct
        else ("second pass over the raw tensors (k_pack_obs)" if tracker is not None else None),
    )
    road_kernel = "k_map_obs_linear" if linear else "k_map_obs+k_map_rows"
    names = {0: "k_world_step", 1: road_kernel}
    if workload == "lidar":
        names[2] = "k_lidar"
    if workload == "bev":
        names[3] = "k_bev"
    kt = {}
    for kid, kname in names.items():
        ms, n = sim.kernel_timing_read(kid)
        kt[kname] = dict(avg_us=1e3 * ms / max(n, 1), launches=n)
    # the partner rows run on the engine's second stream beside the road observation: timed, but not part of the sum that
    # the stretch's wall clock is compared with
    ms, n = sim.kernel_timing_read(4)
    overlapped = {"k_partner_rows": dict(avg_us=1e3 * ms / max(n, 1), launches=n)} if n else {}
    sim.kernel_timing(False)
    # algorithmic bytes per launch, SURVEY.md 8d: per world 36*R_w + 16*N_w + 7200*N_w (road selection + row write-out:
    # k_map_obs hands its selection to k_map_rows; the two launches are one reference system and are timed together)
    alg_bytes = 36.0 * roads + (16.0 + 7200.0) * live
    # ... and what THIS design cannot move less than (a strict lower bound of its HBM traffic): the scan reads an 8-byte
    # (x, y) per road once per world, a world's agents gather at least K distinct 32-byte road records between them, and
    # every live agent's header (16 B) + K rows (7200 B) are written.  `traffic` (PMC) can be below the SURVEY figure
    # (its 36 B per road is not what this layout reads) but never below this one.
    design_min = float(sum(8.0 * int(r) + 32.0 * min(int(r), 200) + 7216.0 * int(n) for n, r in shape))
    survey_bytes = alg_bytes
    scanned = None
    if linear:
        # linear mode: the reference's loop stops at the K-th road in reach, so the scan's algorithmic bytes are 8 B (x, y) per road
        # VISITED, not SURVEY 8d's 36 B for every road of the world; the rows are the same 7216 B per live agent.  SURVEY's form
        # rides along as `survey_bytes_per_launch`.
        # Per WORLD, like SURVEY's form: a world's agents read the same roads, which HBM delivers once -- 8 B x the longest
        # prefix any agent of the world visits.  (8 B x every agent's own prefix is what the L2 serves, not a byte count an HBM
        # roofline can be priced against: 1.5 GB per step on the bench scene.)
        scanned, scanned_world = linear_roads_scanned(sim, kw["observationRadius"])
        alg_bytes = 8.0 * scanned_world + (16.0 + 7200.0) * live
        design_min = alg_bytes
    # Agents whose rows were left in place (pose bits unchanged since they were written: parked and finished agents) moved no
    # bytes: they are taken out of the byte count that `achieved` / `frac` price, so that the fraction stays a statement about
    # bytes the kernel moved; the reference's own count rides along as `reference_bytes_per_launch`.
    launches = max(kt[road_kernel]["launches"], 1)
    skipped_per_launch = rows_skipped / launches
    reference_bytes = alg_bytes
    alg_bytes -= 7216.0 * min(skipped_per_launch, live)
    avg_s = kt[road_kernel]["avg_us"] * 1e-6
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    res["kernels"] = kt
    # HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this same command
    # (tools/profile.sh, committed under profiles/); counters cannot be read from inside the process.  The file names
    # the source stamp of the build it was collected on; another build gets null.
    traffic = None
    try:
        if linear:
            tkey = workload
        elif workload in ("synthetic", "waymo"):
            tkey = ("set_" if knn_order == 1 else "exact_") + workload
        else:  # lidar, bev, rl_loop: collected in the default (reference) row order only
            tkey = workload if knn_order == 0 else ({"cfg3": "set_cfg3", "rl_loop": "rl_loop_set"}.get(workload))
        if agents_override and workload != "ppo_default":
            tkey = (tkey or "") + "_128"
        stamp = source_stamp()
        for tname in sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_traffic.json") and n[:1] == "r"),
                            reverse=True):  # newest round first; only a file collected on THIS build counts
            with open(os.path.join(ROOT, "profiles", tname)) as fh:
                tj = json.load(fh)
            if tj.get("source_stamp") == stamp and args.worlds == (4096 if workload == "cfg3" else 1024):
                traffic = tj.get(tkey, {}).get("hbm_bytes_per_launch")
                break
    except Exception:
        traffic = None
    res["roofline"] = dict(bound="hbm", kernel=road_kernel, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                           traffic_note="(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch, rocprofv3 --pmc passes of this build (profiles/): "
                                        "FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md, which is calibrated for wide "
                                        "coalesced streams -- it OVER-counts the 8- and 16-byte strided / gathered reads of the rank "
                                        "replay's scratch rows and of the road records (the guide calls other widths uncalibrated)",
                           algorithmic_bytes_per_launch=alg_bytes, avg_kernel_us=kt[road_kernel]["avg_us"],
                           design_min_bytes_per_launch=design_min,
                           survey_bytes_per_launch=survey_bytes, reference_bytes_per_launch=reference_bytes,
                           frac_of_reference_bytes=(reference_bytes / avg_s / 1e9 / HBM_PEAK_GBS) if avg_s > 0 else 0.0,
                           roads_visited_until_k=scanned, agents_skipped_per_launch=skipped_per_launch,
                           skip_note="an agent whose pose bits equal the ones its rows were last written for is not rewritten (parked / "
                                     "finished agents); its 7216 B are NOT in algorithmic_bytes_per_launch, which `achieved` and `frac` "
                                     "price (bytes the kernel moved); reference_bytes_per_launch counts every live agent like the "
                                     "reference's loop does, and frac_of_reference_bytes = that / time / peak can exceed what any "
                                     "memory system delivers -- it is the speed-up over rewriting everything, not a bandwidth",
                           denominators="algorithmic_bytes_per_launch = SURVEY 8d's contract (36 R_w + 7216 N_w per world-step); "
                                        "design_min_bytes_per_launch = this layout's own floor (8 R_w scanned once per world + 32 B x "
                                        "min(R_w, K) gathered records + 7216 N_w written): traffic >= the second, not necessarily the first")
    # the other kernels with a SURVEY 8d byte count, same steps
    extra = {}
    if "k_lidar" in kt and kt["k_lidar"]["avg_us"] > 0:
        # returns of agents around whom nothing changed are left in place (k_world_step's flags; gd_stat 44: the last launch's count)
        traced = sim.stat(44)
        b, ref_b = 2400.0 * traced + 36.0 * roads, 2400.0 * live + 36.0 * roads
        t = kt["k_lidar"]["avg_us"] * 1e-6
        extra["k_lidar"] = dict(algorithmic_bytes_per_launch=b, achieved=b / t / 1e9, frac=b / t / 1e9 / HBM_PEAK_GBS, unit="GB/s",
                                agents_traced_last_launch=traced, reference_bytes_per_launch=ref_b,
                                frac_of_reference_bytes=ref_b / t / 1e9 / HBM_PEAK_GBS)
    if "k_bev" in kt and kt["k_bev"]["avg_us"] > 0:
        # rasters whose agent's surroundings did not change are left in place (k_world_step's dirty flags): the bytes moved are
        # those of the rasters painted (gd_stat 31: the last launch's count), the reference rewrites every live agent's
        painted = sim.stat(31)
        b, ref_b = 160000.0 * painted, 160000.0 * live
        t = kt["k_bev"]["avg_us"] * 1e-6
        extra["k_bev"] = dict(algorithmic_bytes_per_launch=b, achieved=b / t / 1e9, frac=b / t / 1e9 / HBM_PEAK_GBS, unit="GB/s",
                              rasters_painted_last_launch=painted, reference_bytes_per_launch=ref_b,
                              frac_of_reference_bytes=ref_b / t / 1e9 / HBM_PEAK_GBS)
    if kt["k_world_step"]["avg_us"] > 0:
        # state (40 + 52 + 28) + self / absolute rows 88 + partner rows 36 (A - 1) bytes per live agent; with the
        # partner rows in a kernel of their own (k_partner_rows) each kernel is priced with its own share
        own = (40.0 + 52.0 + 28.0 + 88.0) * live
        part = 36.0 * (args.agents - 1) * live
        b = own if overlapped else own + part
        extra["k_world_step"] = dict(algorithmic_bytes_per_launch=b, achieved=b / (kt["k_world_step"]["avg_us"] * 1e-6) / 1e9,
                                     frac=b / (kt["k_world_step"]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, unit="GB/s")
        if overlapped and overlapped["k_partner_rows"]["avg_us"] > 0:
            t = overlapped["k_partner_rows"]["avg_us"] * 1e-6
            extra["k_partner_rows"] = dict(algorithmic_bytes_per_launch=part, achieved=part / t / 1e9,
                                           frac=part / t / 1e9 / HBM_PEAK_GBS, unit="GB/s",
                                           note="runs on a second stream beside the road observation")
    res["other_rooflines"] = extra
    # how many of the live agents see any road at all at the end of the run (a finished agent parked at the padding
    # position is out of reach of every road: no selection work, 7200 B of padding rows all the same)
    amap = sim.agent_roadmap_tensor().to_torch()
    res["live_agents_with_roads_in_reach"] = int((amap[..., 6] > 0).any(dim=-1).sum().item())
    res["kernels_sum_us"] = sum(v["avg_us"] for v in kt.values())
    res["overlapped_kernels"] = overlapped
    res["engine"] = dict(graph_steps=sim.stat(0), plain_steps=sim.stat(1), graph_captures=sim.stat(2),
                         set_order_rows_fused=sim.stat(3), set_order_agents_per_wave=sim.stat(4),
                         road_kernel_agents_per_wave=sim.stat(6), rank_audit_violations=sim.stat(21),
                         schedule_env={k: os.environ[k] for k in ("GPUDRIVE_NO_GRAPH", "GPUDRIVE_SET_FUSED_ROWS",
                                                                   "GPUDRIVE_SET_AGENTS_PER_WAVE") if k in os.environ})
    # ---- BASELINE configs[3]: observation all-gather over RCCL, overlapped with the next step ----
    if res["engine"]["rank_audit_violations"] != 0:  # an index of the rank path left its array (clamped on the device): no number of this run counts
        raise SystemExit("bench.py: %s: the rank path's bounds audit counted %d violations (gd_stat 21)" % (name, res["engine"]["rank_audit_violations"]))
    if world > 1 and args.gather != "none" and name == args.workloads.split(",")[0]:
        res["allgather"] = gather_stretch(sim, batches, all_worlds, args, k, device, world)
    sim.close()
    return res


def gather_stretch(sim, batches, all_worlds, args, k, device, world):
    """Every step: simulator step, fused observation pack, all-gather of the packed block (raw) or of the controlled
    agents' rows (compact) on a side stream while the next step runs (sharding.ObservationGather)."""
    import torch.distributed as dist
    ctrl = sim.controlled_state_tensor().to_torch()[..., 0] == 1
    if args.gather_payload == "packed":
        D = 6 + (args.agents - 1) * 6 + 200 * 13
        parts = [(sharding.ObservationGather(args.gather, args.worlds * args.agents, D, device, timing=True), sim.packed_observations)]
    else:  # raw_rows: the exported rows themselves, three gathers per step (no concatenation pass), a quarter fewer bytes
        parts = [(sharding.ObservationGather(args.gather, args.worlds * args.agents, dim, device, timing=True), getter) for dim, getter in (
            (8, lambda: sim.self_observation_tensor().to_torch()),
            ((args.agents - 1) * 9, lambda: sim.partner_observations_tensor().to_torch()),
            (200 * 9, lambda: sim.agent_roadmap_tensor().to_torch()))]
    for g, _ in parts:
        g.set_mask(ctrl)
    og = parts[-1][0]
    act = sim.action_tensor().to_torch()
    steps = args.gather_steps

    def one(i):
        act.copy_(batches[i % len(batches)])
        sim.step()
        blocks = [getter() for _, getter in parts]
        if i > k:
            for g, _ in parts:
                g.wait()       # the learner would consume step i-1's gathered blocks here
        for (g, _), blk in zip(parts, blocks):
            g.start(blk)
    for i in range(k, k + 3):
        one(i)
    for g, _ in parts:
        g.wait()
        g.events.clear()
    sharding.barrier(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(k + 3, k + 3 + steps):
        one(i)
    for g, _ in parts:
        full, counts = g.wait()
    torch.cuda.synchronize(device)
    sharding.barrier(device)
    elapsed = sharding.reduce_max(time.perf_counter() - t0, device)
    ms = sum(g.mean_ms() or 0.0 for g, _ in parts)
    per_rank = sum(g.bytes_per_rank for g, _ in parts)
    share = float(ctrl.float().mean().item())
    return dict(mode=args.gather, payload=args.gather_payload, bytes_per_rank_by_payload_and_mode=gather_bytes_table(args.worlds, args.agents, share),
                backend=dist.get_backend(), world_size=dist.get_world_size(), steps=steps,
                bytes_per_rank=per_rank, rows_per_rank=int(og.cap), controlled_per_rank=[int(c) for c in counts.tolist()],
                allgather_ms=ms, ms_per_step_with_gather=1e3 * elapsed / steps,
                gb_per_s_per_link=(per_rank / (ms * 1e-3) / 1e9) if ms else None,
                gathered_shape=list(full.shape),
                note="per-link rate = one peer's block / gather time (every peer's block rides its own xGMI link)")


def gather_bytes_table(worlds, agents, controlled_share=None):
    """Bytes one rank contributes to config 4's all-gather per step, by payload and mode: `packed` = the learner's normalised
    row (6 + (A - 1) * 6 + 200 * 13 floats per agent slot), `raw_rows` = the exported rows it is made of (self 8 + partner
    (A - 1) * 9 + road 200 * 9 floats: a quarter fewer floats than the one-hot packed form; the learner then normalises after the
    gather); `raw` = every agent slot, `compact` = controlled agents only (share given, else left symbolic as bytes per agent)."""
    per_agent = {"packed": 4 * (6 + (agents - 1) * 6 + 200 * 13), "raw_rows": 4 * (8 + (agents - 1) * 9 + 200 * 9)}
    out = {}
    for payload, b in per_agent.items():
        out[payload] = dict(bytes_per_agent=b, raw=b * worlds * agents,
                            compact=(int(b * worlds * agents * controlled_share) if controlled_share is not None else None))
    return out


def cpu_baseline(args, budget_s=15.0):
    """The oracle (a port of the reference's CPU algorithm: per-agent heap k-NN, per-pair OBB, AoS,
    one world per task across all host cores like Madrona's ThreadPoolExecutor) timed on this box's
    host cores on a bounded sample of the primary workload."""
    from oracle import oracle as O
    so = None
    try:  # host-tuned build for a fair CPU number; falls back to the portable build
        so = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getpid())
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-fopenmp", "-shared",
                               "-o", so, os.path.join(ROOT, "oracle", "gd_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        so = None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kw = params_for("synthetic")
    rng = np.random.default_rng(0)

    def run(worlds, threads, budget):
        os.environ["OMP_NUM_THREADS"] = str(threads)
        scenes = scenes_for("synthetic", worlds, 0)
        sim = O.OracleSim(scenes, O.default_params(**kw), max_agents=args.agents, lib_path=so)
        if hasattr(sim.L, "orc_set_threads"):
            sim.L.orc_set_threads(int(threads))
        live = int(sim.shape_tensor()[:, 0].sum())
        act = sim.action_tensor()
        # the actions are drawn BEFORE the clock starts, like the GPU side's resident batches: the timed loop is a copy and step()
        draws = []
        for _ in range(8):
            a = np.zeros_like(act)
            a[..., 0] = rng.uniform(-3, 2, act.shape[:2])
            a[..., 1] = rng.uniform(-0.7, 0.7, act.shape[:2])
            draws.append(a)

        def one(k):
            np.copyto(act, draws[k % len(draws)])
            sim.step()
        t0 = time.perf_counter()
        one(0)
        first = time.perf_counter() - t0
        steps = int(max(3, min(200, budget / max(first, 1e-4))))
        t0 = time.perf_counter()
        for k in range(steps):
            one(k + 1)
        dt = time.perf_counter() - t0
        sim.close()
        return live * steps / dt, worlds, steps, dt
    # one world per thread (Madrona's ThreadPoolExecutor runs one world per task, src/mgr.cpp:527-535).  The headline CPU
    # figure uses at most 64 threads; the same with EVERY host core is measured beside it (`all_cores_value`: about half the
    # 64-thread rate on the 256-hardware-thread boxes of this pool -- reported as measured; what bounds it there has not been
    # looked into, and it is a stated baseline, not a target).
    threads = max(1, min(cores, 64))
    rate, worlds, steps, dt = run(threads, threads, budget_s * 0.5)
    rate_all, steps_all, dt_all = None, 0, 0.0
    if cores > threads:
        rate_all, _, steps_all, dt_all = run(cores, cores, budget_s * 0.3)
    rate1, _, steps1, dt1 = run(1, 1, budget_s * 0.2)
    return dict(value=rate, unit="agent-steps/s", cores=threads, host_cores=cores, kind="port",
                single_thread_value=rate1, all_cores_value=rate_all,
                sample="%d synthetic exact-64 worlds (R_w=4096) x %d steps on %d OpenMP threads (one world per thread), %.1f s; "
                       "%s1 world x %d steps on 1 thread, %.1f s; the reference's own CPU ExecMode cannot be built "
                       "(Madrona submodule absent)"
                       % (worlds, steps, threads, dt,
                          ("%d worlds x %d steps on all %d host cores, %.1f s; " % (cores, steps_all, cores, dt_all)) if rate_all else "",
                          steps1, dt1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=273)
    ap.add_argument("--warmup", type=int, default=91)
    ap.add_argument("--worlds", type=int, default=1024, help="worlds per GPU")
    ap.add_argument("--agents", type=int, default=64, choices=(64, 128))
    ap.add_argument("--roofline-steps", type=int, default=0, help="ignored (kept for old command lines): kernels are timed over the timed steps themselves")
    ap.add_argument("--gather", default="compact", choices=("none", "raw", "compact"),
                    help="N > 1 only: after the timed region, a stretch with the observation all-gather of BASELINE configs[3] "
                         "(raw = every agent slot's packed observation, compact = controlled agents only), overlapped with the next step")
    ap.add_argument("--gather-payload", default="packed", choices=("packed", "raw_rows"),
                    help="what the all-gather stretch sends: the packed, normalised observation (6 + (A - 1) * 6 + 200 * 13 floats per "
                         "agent) or the exported self / partner / road rows it is made of (8 + (A - 1) * 9 + 200 * 9 floats: a quarter fewer bytes)")
    ap.add_argument("--gather-steps", type=int, default=30)
    ap.add_argument("--workloads", default="synthetic,waymo,cfg3,lidar,bev,rl_loop,synthetic_128,waymo_raw,synthetic_set,waymo_set,cfg3_set,"
                                            "synthetic_linear,waymo_linear,ppo_default,rl_loop_set",
                    help="first = primary; synthetic | waymo | cfg3 | lidar (Waymo tiles + 360-degree LiDAR) | bev | rl_loop | "
                         "waymo_raw (the Waymo tiles with unreduced polylines, the setting of the reference's C++ tests); "
                         "a _128 suffix = 128 agent slots per world (the fork's kMaxAgentCount), a _set suffix = the same scenes "
                         "in set order (knn_order 1), a _linear suffix = the same scenes with the linear road selection (the "
                         "reference's EnvConfig default); ppo_default = what the reference's PPO baselines construct (Waymo tiles, "
                         "128 agent slots, linear, init_mode all_non_trivial, vehicles only)")
    ap.add_argument("--spin-ms", type=float, default=200.0,
                    help="untimed steps for this many ms of wall clock before each timed stretch (device spin-up after the idle "
                         "world build; 0 = none)")
    ap.add_argument("--no-align", action="store_true",
                    help="do not run on to the first step of an episode before a timed stretch (profiler passes: their counter "
                         "means would otherwise cover up to 90 alignment steps per stretch)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank joins the process group, the ranks are counted with one all-reduce and rank 0 "
                         "prints a line with `distributed` filled in (checks the launch path, e.g. on a CPU-only box with "
                         "--dist-backend gloo)")
    ap.add_argument("--dry-run-sleep", type=float, default=0.0, help="--dry-run only: every rank sleeps this long before it leaves (tests of the launch path's signal handling)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="--dry-run only: this rank leaves with exit code 3 after the ranks were counted (tests of the launch path's exit code, with or without a GPU)")
    ap.add_argument("--headless", action="store_true",
                    help="also print the reference CLI's two lines (src/headless.cpp:145-155) for the primary workload on stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default=None, choices=(None, "nccl", "gloo"),
                    help="default: nccl (RCCL); gloo + --single-device rehearses the N>1 path on one GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--knn-order", type=int, default=0, choices=(0, 1),
                    help="0 = reference heap order (default, elementwise parity); 1 = same row set, in the engine's own (grid-cell) order")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as CHILD processes
    # (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and BEFORE anything here touches the GPU -- a
    # process that has initialised HIP must never be replaced or forked.  The children print rank 0's JSON line on our
    # stdout; our exit code is theirs.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # --standalone: the launcher picks its own free rendezvous port (no bind-then-close window in which another process can
        # take it); --local-addr: the container's hostname may not resolve.  The children get a session of their own so that a
        # signal that ends this process (a `timeout` around it, Ctrl-C) ends every rank with it instead of leaving them on the
        # GPUs; our exit code is theirs.
        import signal
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--standalone", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without a launcher: spawning %s\n" % (args.gpus, " ".join(cmd[1:9])))
        child = subprocess.Popen(cmd, start_new_session=True)

        def end_children(signum=None, frame=None):
            if child.poll() is None:
                try:
                    os.killpg(child.pid, signal.SIGTERM)
                    try:
                        child.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            if signum is not None:
                raise SystemExit(128 + signum)
        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            signal.signal(sig, end_children)
        try:
            rc = child.wait()
        finally:
            end_children()
        raise SystemExit(rc)

    if args.dry_run:
        rank, local_rank, world = sharding.init_process_group(backend=args.dist_backend or "gloo")
        if world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        import torch.distributed as dist
        seen = torch.ones(1)
        gather = None
        if dist.is_initialized():
            dist.all_reduce(seen)
            # one round of config 4's observation gather on host tensors of the real row shape (2 worlds per rank), both modes:
            # rank r's rows must land in section r on every rank
            W, A = 2, args.agents
            D = 6 + (A - 1) * 6 + 200 * 13
            g = torch.Generator().manual_seed(7 + rank)
            block = torch.full((W, A, D), float(rank))
            ctrl = torch.rand(W, A, generator=g) < 0.25 + 0.05 * rank   # unequal controlled counts
            ok = True
            for mode in ("raw", "compact"):
                og = sharding.ObservationGather(mode, W * A, D, torch.device("cpu"))
                og.set_mask(ctrl)
                og.start(block)
                full, counts = og.wait()
                for r in range(world):
                    n = int(counts[r])
                    ok = ok and n > 0 and bool((full[r * og.cap:r * og.cap + n] == float(r)).all())
            gather = dict(ok=bool(ok), rows_per_rank_compact=int(og.cap), controlled_per_rank=[int(c) for c in og.counts.tolist()])
            if args.dry_run_sleep > 0:
                time.sleep(args.dry_run_sleep)
        if rank == args.dry_run_fail_rank:
            raise SystemExit(3)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_counted": int(seen.item()), "gather_round": gather,
                              "gather_bytes_per_rank": gather_bytes_table(args.worlds, args.agents),
                              "distributed": dict(world_size=dist.get_world_size(), backend=dist.get_backend())
                              if dist.is_initialized() else None}), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the step path has no CPU fallback")
    rank, local_rank, world = sharding.init_process_group(backend=args.dist_backend)
    if args.single_device:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    results = [bench_workload(w, args, rank, local_rank, world, device) for w in args.workloads.split(",")]
    primary = results[0]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    dist_info = None
    if torch.distributed.is_initialized():
        import torch.distributed as dist
        dist_info = dict(world_size=dist.get_world_size(), backend=dist.get_backend(),
                         ms_per_step_max_rank=primary["ms_per_step"], ms_per_step_min_rank=primary["ms_per_step_min_rank"])
    if rank == 0:
        order_txt = {0: "reference heap order", 1: "SET order: same rows as a set, grid-cell order"}[primary["knn_order"]]
        line = {
            "metric": "agent-steps/sec at 1024 worlds x 64 agents; achieved HBM GB/s on obs kernel",
            "value": primary["agent_steps_per_s"],
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": primary["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%d worlds x %d agents per GPU, classic bicycle dynamics, %d-partner + 200-road-point "
                            "k-NN obs (%s), radius 50, collisions ignored, reset every 91 steps; "
                            "scenes: seeded synthetic exact-64 (64 live agents, 4096 road-edge segments per world)"
                            % (args.worlds, args.agents, args.agents - 1, order_txt),
                "worlds_per_gpu": args.worlds, "max_agents": args.agents,
                "parallelism": "worlds sharded %d-way, no per-step collective" % world,
            },
            "padded_agent_steps_per_s": primary["padded_agent_steps_per_s"],
            "timed_region": "step() as a user calls it: the step's kernels replayed from the captured hipGraph (%d graph / %d plain "
                            "steps), no instrumentation" % (primary["timed_region"]["graph_steps"], primary["timed_region"]["plain_steps"]),
            "ms_per_step_events": primary["ms_per_step_events"],
            "gc_ms_in_timed_stretches": primary["gc_ms_in_timed_stretches"],
            "spin_up": "%g ms of untimed steps before each timed stretch (%d steps on the primary workload): after the idle world "
                       "build the GPU delays its first work by 10-90 ms in about one stretch of twelve" % (args.spin_ms, primary["spin_up_steps"]),
            "python_gc": "left alone (GPUDRIVE_BENCH_KEEP_GC=1)" if KEEP_GC else "collected before and disabled during each timed stretch",
            "kernels_sum_us": primary["kernels_sum_us"],
            "roofline": primary["roofline"],
            "kernels": primary["kernels"],
            "overlapped_kernels": primary["overlapped_kernels"],
            "kernel_times": "HIP events around every launch of a second stretch of the same K steps right after the timed region "
                            "(kernels launched one by one; its wall clock is ms_per_step_events)",
            "other_rooflines": primary.get("other_rooflines"),
            "engine": dict(primary.get("engine", {}), source_stamp=source_stamp()),
            "distributed": dist_info,
            "allgather": primary.get("allgather"),
            "cpu_baseline": cpu,
            "other_workloads": [
                {k: r[k] for k in ("workload", "knn_order", "worlds", "packed_observation", "agent_steps_per_s", "padded_agent_steps_per_s", "ms_per_step",
                                   "ms_per_step_events", "gc_ms_in_timed_stretches", "spin_up_steps", "overlapped_kernels", "kernels_sum_us", "live_agents_per_rank", "live_agents_with_roads_in_reach", "road_entities_per_rank",
                                   "roofline", "kernels", "other_rooflines")}
                for r in results[1:]],
            "init_seconds": primary["init_seconds"],
            "live_agents_per_rank": primary["live_agents_per_rank"],
            "live_agents_with_roads_in_reach": primary["live_agents_with_roads_in_reach"],
        }
        print(json.dumps(line), flush=True)
        if args.headless:
            # the reference CLI's two lines (src/headless.cpp:145-155): FPS = steps * worlds / s, and its "Agent-Normalized
            # FPS" = FPS * total live agents (as printed there -- not divided by the worlds); then what it presumably means
            fps = args.steps * args.worlds * world / primary["seconds"]
            tot = primary["live_agents_per_rank"] * world
            sys.stderr.write("FPS %f\nAgent-Normalized FPS %f\n" % (fps, fps * tot))
            sys.stderr.write("(live agent-steps per second: %f)\n" % primary["agent_steps_per_s"])
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

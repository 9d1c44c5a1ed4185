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
"""
import argparse
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
EPISODE = 91

WAYMO = [os.path.join(ROOT, "tests", "data", n) for n in
         ("test.json", "tfrecord-00002-of-01000_407.json", "tfrecord-00000-of-01000_4.json")]


def params_for(workload):
    kw = dict(observationRadius=50.0, collisionBehaviour=2, rewardType=1, distanceToGoalThreshold=2.0,
              dynamicsModel=0, roadObservationAlgorithm=0, isStaticAgentControlled=1,
              initOnlyValidAgentsAtFirstStep=0, IgnoreNonVehicles=0)
    kw["polylineReductionThreshold"] = 0.0 if workload in ("synthetic", "rl_loop") else 0.1
    if workload == "lidar":  # BASELINE configs[4]: LiDAR 3 x 50 rays, mixed vehicle / cyclist / pedestrian agents
        kw["enableLidar"] = 1
    if workload == "cfg3":   # BASELINE configs[2]: 4 x the worlds, collisions stop agents, goal-reach reward
        kw["collisionBehaviour"] = 0
    return kw


def scenes_for(workload, worlds, rank):
    if workload in ("synthetic", "rl_loop"):
        d = os.path.join(tempfile.gettempdir(), "gpudrive_amd_bench_scenes")
        paths = synth.write_scenes(d, [rank * 1000 + i for i in range(8)])
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


def bench_workload(workload, args, rank, local_rank, world, device):
    # everything (action writes, steps, resets) runs on one side stream: the engine replays its step as
    # a hipGraph there (the legacy null stream cannot be captured)
    with torch.cuda.stream(torch.cuda.Stream(device=device)):
        return _bench_workload(workload, args, rank, local_rank, world, device)


def _bench_workload(workload, args, rank, local_rank, world, device):
    kw = params_for(workload)
    if workload == "cfg3":
        args = argparse.Namespace(**dict(vars(args), worlds=4 * args.worlds))
    scenes = scenes_for(workload, args.worlds, rank)
    t0 = time.time()
    sim = make_sim(scenes, kw, args.agents, local_rank, knn_order=args.knn_order,
                   lidar_half_angle=float(np.pi) if workload == "lidar" else 0.0,  # 360 degrees
                   enable_bev=workload == "bev")  # the reference rasterises the 200 x 200 BEV on every step
    torch.cuda.synchronize(device)
    init_s = time.time() - t0
    shape = sim.shape_tensor().to_torch().cpu().numpy()
    live = int(shape[:, 0].sum())
    roads = int(shape[:, 1].sum())
    batches = action_batches(args.worlds, args.agents, device, seed=1234 + rank)
    all_worlds = np.arange(args.worlds, dtype=np.int32)
    tracker = None
    if workload == "rl_loop":
        from gpudrive_lab_amd.episode import EpisodeTracker
        tracker = EpisodeTracker(sim)

    k = run_steps(sim, batches, all_worlds, args.warmup, tracker=tracker)
    sharding.barrier(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    k = run_steps(sim, batches, all_worlds, args.steps, start=k, tracker=tracker)
    torch.cuda.synchronize(device)
    sharding.barrier(device)
    elapsed = time.perf_counter() - t0
    elapsed = sharding.reduce_max(elapsed, device)
    total_live = sharding.reduce_sum(live, device)
    res = dict(
        workload=workload, seconds=elapsed, ms_per_step=1e3 * elapsed / args.steps,
        live_agents_per_rank=live, road_entities_per_rank=roads, init_seconds=init_s,
        agent_steps_per_s=total_live * args.steps / elapsed,
        padded_agent_steps_per_s=world * args.worlds * args.agents * args.steps / elapsed,
    )
    # dominant-kernel roofline: HIP events on the engine's stream around every launch of the road
    # observation kernel, over a separate timed stretch (so `value` is not perturbed)
    sim.kernel_timing(True)
    run_steps(sim, batches, all_worlds, args.roofline_steps, start=k, tracker=tracker)
    torch.cuda.synchronize(device)
    names = {0: "k_world_step", 1: "k_map_obs"}
    if workload == "lidar":
        names[2] = "k_lidar"
    if workload == "bev":
        names[3] = "k_bev"
    kt = {}
    for kid, name in names.items():
        ms, n = sim.kernel_timing_read(kid)
        kt[name] = dict(avg_us=1e3 * ms / max(n, 1), launches=n)
    sim.kernel_timing(False)
    # algorithmic bytes per launch, SURVEY.md 8d: per world 36*R_w + 16*N_w + 7200*N_w
    alg_bytes = 36.0 * roads + (16.0 + 7200.0) * live
    avg_s = kt["k_map_obs"]["avg_us"] * 1e-6
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    res["kernels"] = kt
    # HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this same command
    # (tools/profile.sh, committed under profiles/); counters cannot be read from inside the process.
    traffic = None
    try:
        tkey = ("set_" if args.knn_order == 1 else "exact_") + workload if workload != "lidar" else "lidar"
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            traffic = json.load(fh).get(tkey, {}).get("hbm_bytes_per_launch")
        if args.worlds != 1024 or args.agents != 64:
            traffic = None
    except Exception:
        traffic = None
    res["roofline"] = dict(bound="hbm", kernel="k_map_obs", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                           algorithmic_bytes_per_launch=alg_bytes, avg_kernel_us=kt["k_map_obs"]["avg_us"])
    sim.close()
    return res


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
    worlds = max(cores * 2, 16)
    kw = params_for("synthetic")
    scenes = scenes_for("synthetic", worlds, 0)
    sim = O.OracleSim(scenes, O.default_params(**kw), max_agents=args.agents, lib_path=so)
    live = int(sim.shape_tensor()[:, 0].sum())
    rng = np.random.default_rng(0)
    act = sim.action_tensor()

    def one():
        act[..., 0] = rng.uniform(-3, 2, act.shape[:2])
        act[..., 1] = rng.uniform(-0.7, 0.7, act.shape[:2])
        sim.step()
    t0 = time.perf_counter()
    one()
    first = time.perf_counter() - t0
    steps = int(max(3, min(200, budget_s / max(first, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    sim.close()
    return dict(value=live * steps / dt, unit="agent-steps/s", cores=cores, kind="port",
                sample="%d synthetic exact-64 worlds (R_w=4096) x %d steps, OpenMP over worlds, %.1f s; "
                       "the reference's own CPU ExecMode cannot be built (Madrona submodule absent)"
                       % (worlds, steps, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=273)
    ap.add_argument("--warmup", type=int, default=91)
    ap.add_argument("--worlds", type=int, default=1024, help="worlds per GPU")
    ap.add_argument("--agents", type=int, default=64, choices=(64, 128))
    ap.add_argument("--roofline-steps", type=int, default=91, help="separately timed stretch for per-kernel HIP-event timing: one whole episode")
    ap.add_argument("--workloads", default="synthetic,waymo,cfg3,lidar,bev,rl_loop",
                    help="first = primary; synthetic | waymo | lidar (Waymo tiles + 360-degree LiDAR)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default=None, choices=(None, "nccl", "gloo"),
                    help="default: nccl (RCCL); gloo + --single-device rehearses the N>1 path on one GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--knn-order", type=int, default=0, choices=(0, 1),
                    help="0 = reference heap order (default, elementwise parity); 1 = same row set, road-index order")
    args = ap.parse_args()

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
    if rank == 0:
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
                            % (args.worlds, args.agents, args.agents - 1,
                               "reference heap order" if args.knn_order == 0 else "SET order: same rows, road-index order"),
                "worlds_per_gpu": args.worlds, "max_agents": args.agents,
                "parallelism": "worlds sharded %d-way, no per-step collective" % world,
            },
            "padded_agent_steps_per_s": primary["padded_agent_steps_per_s"],
            "roofline": primary["roofline"],
            "kernels": primary["kernels"],
            "cpu_baseline": cpu,
            "other_workloads": [
                {k: r[k] for k in ("workload", "agent_steps_per_s", "padded_agent_steps_per_s", "ms_per_step",
                                   "live_agents_per_rank", "road_entities_per_rank", "roofline", "kernels")}
                for r in results[1:]],
            "init_seconds": primary["init_seconds"],
        }
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

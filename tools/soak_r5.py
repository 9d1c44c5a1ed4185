#!/usr/bin/env python3
"""developer tool, round 5: the two "leave it where it is" mechanisms at bench size, for hundreds of steps with whole-batch and
partial resets -- (a) pose stamps: a simulator that skips rows against one that rewrites everything (GPUDRIVE_NO_POSE_SKIP=1),
agent_roadmap_tensor compared BITWISE (mode "bev": the same twin with BEV rasters attached, the rasters compared; mode "lidar": the LiDAR returns); (b) the packed observation written by the step (pack only) against the second pass over
the raw tensors of a twin simulator, compared bitwise.  gpurun -- python tools/soak_r5.py [steps]   (output kept as
profiles/r05_soak.txt)"""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)


def twin(wl, W, env):
    name, order, agents = bench.split_workload(wl)
    agents = agents or 64
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sim = bench.make_sim(bench.scenes_for(name, W, 0, agents), bench.params_for(name), agents, 0, knn_order=order, enable_bev=(name == "bev"),
                             lidar_half_angle=float(np.pi) if name == "lidar" else 0.0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return sim, agents


def run(wl, W, mode):
    t0 = time.time()
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        a, agents = twin(wl, W, {})
        b, _ = twin(wl, W, {"GPUDRIVE_NO_POSE_SKIP": "1"} if mode in ("skip", "bev", "lidar") else {})
        if mode == "pack":
            assert a.direct_pack(only=True)
        batches = bench.action_batches(W, agents, dev, seed=77)
        rng = np.random.default_rng(3)
        checks = bad = 0
        a.stat(30)
        for k in range(STEPS):
            for s in (a, b):
                s.action_tensor().to_torch().copy_(batches[k % 8])
                s.step()
            if k % 91 == 90:
                for s in (a, b):
                    s.reset(list(range(W)))
            elif k % 37 == 36:
                idx = sorted(rng.choice(W, size=W // 8, replace=False).tolist())
                for s in (a, b):
                    s.reset(idx)
            if k % 25 == 24 or k == STEPS - 1:
                if mode == "bev":
                    x, y = a.bev_observation_tensor().to_torch(), b.bev_observation_tensor().to_torch()
                elif mode == "lidar":
                    x, y = a.lidar_tensor().to_torch(), b.lidar_tensor().to_torch()
                elif mode == "skip":
                    x, y = a.agent_roadmap_tensor().to_torch(), b.agent_roadmap_tensor().to_torch()
                else:
                    x, y = a.packed_observations(), b.packed_observations()
                checks += 1
                if not torch.equal(x.view(torch.int32), y.view(torch.int32)):
                    bad += 1
                    print("  %s %s: DIFFERENT at step %d (%d elements)" % (wl, mode, k + 1, int((x.view(torch.int32) != y.view(torch.int32)).sum())))
        torch.cuda.synchronize()
        skipped = a.stat(30)
        print("%-18s %-5s %5d worlds %4d steps %5.1f s: %d bitwise comparisons, %d different; %.0f agents left in place per step; audit %d"
              % (wl, mode, W, STEPS, time.time() - t0, checks, bad, skipped / STEPS, a.stat(21)))
        a.close(); b.close()
    return bad


total = 0
for wl, W, mode in (("ppo_default", 1024, "skip"), ("waymo_linear", 1024, "skip"), ("synthetic_linear", 1024, "skip"), ("waymo_set", 1024, "skip"),
                    ("cfg3", 4096, "skip"), ("waymo", 1024, "skip"), ("synthetic_set", 1024, "pack"), ("synthetic", 1024, "pack"),
                    ("ppo_default", 1024, "pack"), ("waymo_set", 1024, "pack"), ("bev", 256, "bev"), ("lidar", 1024, "lidar")):
    total += run(wl, W, mode)
print("TOTAL different:", total)
sys.exit(1 if total else 0)

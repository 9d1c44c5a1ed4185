#!/usr/bin/env python3
"""developer tool: the product library stepped for many episodes on the bench workloads with resets and map changes in
between; prints the bounds-audit counter of the rank path (gd_stat 21, must stay 0), how the agents were selected and
whether every observation stayed finite.  gpurun -- python tools/soak.py [steps]   (the output is kept as profiles/r04_soak.txt)"""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
for wl, W in (("synthetic", 1024), ("waymo", 1024), ("cfg3", 4096), ("waymo_raw", 1024), ("synthetic_set", 1024), ("waymo_set", 1024), ("cfg3_set", 4096)):
    name, order, agents = bench.split_workload(wl)
    t0 = time.time()
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        sim = bench.make_sim(bench.scenes_for(name, W, 0, agents or 64), bench.params_for(name), agents or 64, 0, knn_order=order)
        batches = bench.action_batches(W, agents or 64, dev, seed=99)
        act = sim.action_tensor().to_torch()
        obs = sim.agent_roadmap_tensor().to_torch()
        rng = np.random.default_rng(5)
        worst = 0
        for k in range(STEPS):
            act.copy_(batches[k % 8]); sim.step()
            if k % 91 == 90:
                sim.reset(list(range(W)))                       # every world, as the bench does
            elif k % 37 == 36:
                sim.reset(sorted(rng.choice(W, size=W // 8, replace=False).tolist()))  # an eighth of them, mid-episode
            if k % 50 == 49:
                torch.cuda.synchronize()
                worst = max(worst, sim.stat(21))
                if not bool(torch.isfinite(obs).all()): print("  NON-FINITE road observation at step", k + 1)
        torch.cuda.synchronize()
        path = sim.debug_road_path()
        print("%-10s %5d worlds %4d steps %5.1f s  audit counter %d  (last step: %d agents by the rank replay, most candidates %d; %d by the history replay; %d other slots)"
              % (wl, W, STEPS, time.time() - t0, max(worst, sim.stat(21)), int((path > 0).sum()), int(path.max()), int((path == -1).sum()),
                 int(((path <= 0) & (path != -1)).sum())))
        sim.close()

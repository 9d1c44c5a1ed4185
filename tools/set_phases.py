#!/usr/bin/env python3
"""developer tool (clock build: tools/build_expt.sh clk -DGD_CLOCKS): how the phases of the set-order selection
(k_map_obs_set) share the waves' time, with the gather iterations and candidates per agent.  gpurun -- python tools/set_phases.py [workload]"""
import os, sys
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GPUDRIVE_DEV"] = "1"
os.environ["GPUDRIVE_AMD_LIB"] = os.path.join(ROOT, "build", "expt", "expt_%s.so" % os.environ.get("EXPT", "clk"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
WL = sys.argv[1] if len(sys.argv) > 1 else "synthetic_set"
name, order, agents = bench.split_workload(WL)
W = 4096 if name == "cfg3" else 1024
dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for(name, W, 0, agents or 64), bench.params_for(name), agents or 64, 0, knn_order=order)
    batches = bench.action_batches(W, agents or 64, dev, seed=1234)
    act = sim.action_tensor().to_torch()
    for k in range(40):
        act.copy_(batches[k % 8]); sim.step()
        if k % 10 == 9:
            torch.cuda.synchronize()
            c = np.array([sim.stat(22 + q) for q in range(8)], np.float64)
            live = sim.stat(5) * 10
            print("steps %2d-%2d  phases %% (prologue, gather, keys+histogram, K-th key, ties+write-out, header): %s   per agent: %.0f ticks, %.1f gather iterations, %.0f candidates"
                  % (k - 8, k + 1, np.round(100 * c[:6] / max(c[:6].sum(), 1), 1), c[:6].sum() / max(live, 1), c[6] / max(live, 1), c[7] / max(live, 1)))
    sim.close()

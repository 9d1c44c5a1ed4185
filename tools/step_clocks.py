#!/usr/bin/env python3
"""developer tool (clock build: tools/build_expt.sh clk -DGD_CLOCKS): how the phases of k_world_step share a workgroup's time
(s_memtime ticks seen by thread 0, averaged over the workgroups).  gpurun -- python tools/step_clocks.py [workload ...]"""
import os, sys
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GPUDRIVE_DEV"] = "1"
os.environ["GPUDRIVE_AMD_LIB"] = os.path.join(ROOT, "build", "expt", "expt_%s.so" % os.environ.get("EXPT", "clk"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
NAMES = ("loads", "movement", "publish+OBB", "agent pairs", "road boxes", "flags..self/abs rows", "partner rows")
dev = torch.device("cuda", 0)
for WL in (sys.argv[1:] or ["synthetic_set"]):
    name, order, agents = bench.split_workload(WL)
    W = 4096 if name == "cfg3" else 1024
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        sim = bench.make_sim(bench.scenes_for(name, W, 0, agents or 64), bench.params_for(name), agents or 64, 0, knn_order=order)
        batches = bench.action_batches(W, agents or 64, dev, seed=1234)
        act = sim.action_tensor().to_torch()
        for k in range(30):
            act.copy_(batches[k % 8]); sim.step()
            if k == 9:
                torch.cuda.synchronize(); sim.stat(32)
        torch.cuda.synchronize()
        c = np.array([sim.stat(32 + q) for q in range(12)], np.float64)
        per = c[:7] / max(c[7], 1)
        print("%-18s ticks per workgroup %6.0f: %s" % (WL, per.sum(), ", ".join("%s %.0f (%.0f %%)" % (n, v, 100 * v / per.sum()) for n, v in zip(NAMES, per))))
        print("   road-box items per workgroup %.0f, past the cull %.0f" % (c[8] / max(c[7], 1), c[9] / max(c[7], 1)))
        sim.close()

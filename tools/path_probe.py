#!/usr/bin/env python3
"""developer tool: which road-selection path every agent took, step by step.  tools/path_probe.py <workload> [worlds]"""
import os, sys, collections
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
name = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
EVERY = int(sys.argv[4]) if len(sys.argv) > 4 else 1
workload, order, agents = bench.split_workload(name)
agents = agents or 64
dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for(workload, W, 0, agents), bench.params_for(workload), agents, 0, knn_order=order)
    batches = bench.action_batches(W, agents, dev, seed=1234)
    act = sim.action_tensor().to_torch()
    for k in range(STEPS):
        act.copy_(batches[k % 8]); sim.step(); torch.cuda.synchronize()
        if (k + 1) % EVERY: continue
        path = sim.debug_road_path()
        c = collections.Counter()
        c["ranked"] = int((path > 0).sum()); c["far"] = int((path == -3).sum()); c["padding"] = int((path == 0).sum())
        for v in (-1, -10, -11, -12, -13, -2):
            c[{-1: "fallback(group)", -10: "no checkpoints/small world", -11: "overflow", -12: "ties>32", -13: "bypass", -2: "rank path off"}[v]] = int((path == v).sum())
        print("step %d:" % (k + 1), dict(c), "max n", int(path.max()), "audit", sim.stat(21))
    sim.close()

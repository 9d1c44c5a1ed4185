#!/usr/bin/env python3
"""developer tool: candidate counts and fallback rate of the rank replay on the bench scenes.  gpurun -- python3 tools/rank_stats.py [synthetic|waymo]"""
import os, sys
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
W = 256
dev = torch.device("cuda", 0)
sim = bench.make_sim(bench.scenes_for(wl, W, 0), bench.params_for(wl), 64, 0)
batches = bench.action_batches(W, 64, dev, seed=1234)
act = sim.action_tensor().to_torch()
for k in range(40):
    act.copy_(batches[k % 8]); sim.step()
    if k in (0, 1, 2, 5, 10, 20, 39):
        p = sim.debug_road_path()
        live = p != 0
        fb = (p == -1) | (p <= -10)
        far = (p == -3)
        import collections
        why = collections.Counter(p[p <= -10].tolist())
        n = p[p > 0]
        # agents that individually overflowed: fallback groups are 32 wide
        print("step %2d: far %d reasons %s" % (k + 1, far.sum(), dict(why)), end=" ")
        print("step %2d: live %d, fallback agents %d (%.1f%%), rank agents %d; candidates mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" %
              (k + 1, live.sum(), fb.sum(), 100.0 * fb.sum() / max(live.sum(), 1), (p > 0).sum(),
               n.mean() if n.size else 0, *(np.percentile(n, [50, 90, 99]) if n.size else (0, 0, 0)), n.max() if n.size else 0))
sim.close()

#!/usr/bin/env python3
"""developer tool (diagnostic build: tools/build_expt.sh diag -DGD_DIAG): per step of the 4096-world config, the road
observation's time next to the most crowded ranking bucket of k_knn_rank and how its phases share the waves' time."""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
os.environ["GPUDRIVE_RANK_DBG"] = "9"  # switches the counters on
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GPUDRIVE_DEV"] = "1"
os.environ["GPUDRIVE_AMD_LIB"] = os.path.join(ROOT, "build", "expt", "expt_%s.so" % os.environ.get("EXPT", "diag"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
WL = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
W = 4096 if WL == "cfg3" else 1024
dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for(WL, W, 0), bench.params_for(WL), 64, 0)
    batches = bench.action_batches(W, 64, dev, seed=1234)
    act = sim.action_tensor().to_torch()
    sim.kernel_timing(True)
    for k in range(50):
        act.copy_(batches[k % 8]); sim.step()
        torch.cuda.synchronize()
        ms, n = sim.kernel_timing_read(1)
        sim.kernel_timing(True)
        path = sim.debug_road_path()
        if os.environ.get('EXPT') == 'clk':
            longest, cand, ins = sim.stat(18), sim.stat(19), sim.stat(20)
            print("        replay: %d candidates beyond K, %d inserts (%.1f %% fail), longest wave %d rounds" % (cand, ins, 100 - 100 * ins / max(cand, 1), longest))
        ph = np.array([sim.stat(10 + q) for q in range(7)], np.float64) if os.environ.get('EXPT') == 'clk' else np.zeros(7)
        print("step %2d road obs %.0f us  widest bucket %d  ranked %d fallback %d far %d max n %d  k_knn_rank phases %% (between agents, words, keys, "
              "count+prefix, scatter, order in buckets, write-out): %s" %
              (k + 1, 1e3 * ms / max(n, 1), sim.stat(8) if os.environ.get('EXPT', 'diag') == 'diag' else -1, (path > 0).sum(), ((path == -1) | (path <= -10)).sum(),
               (path == -3).sum(), path.max(), np.round(100 * ph / max(ph.sum(), 1), 1)))
        if os.environ.get('EXPT') == 'clk':
            print("        trips through the word-expansion loop per ranked agent: %.1f" % (sim.stat(17) / max((path > 0).sum(), 1)))
    sim.close()

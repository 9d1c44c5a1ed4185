#!/usr/bin/env python3
"""developer tool: where does the sporadic 10-80 ms stall inside a 20-step stretch sit -- in a host call or on the GPU?
Per step: host time of the calls, and a GPU event after the step.   gpurun -- python3 tools/stall_probe.py"""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

dev = torch.device("cuda", 0)
args = bench.argparse.Namespace(worlds=1024, agents=64, knn_order=0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for("waymo", 1024, 0), bench.params_for("waymo"), 64, 0)
    batches = bench.action_batches(1024, 64, dev, seed=1)
    act = sim.action_tensor().to_torch()
    mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
    if mode == "events":
        sim.kernel_timing(True)
    for _ in range(5):
        act.copy_(batches[0]); sim.step()
    torch.cuda.synchronize()
    bad = 0
    tot = []
    for rnd in range(150):
        if mode == "sleep":
            time.sleep(0.05)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        host = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(20):
            a = time.perf_counter()
            act.copy_(batches[k % 8])
            b = time.perf_counter()
            sim.step()
            c = time.perf_counter()
            ev[k + 1].record()
            host.append((b - a, c - b, time.perf_counter() - c))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tot.append(el)
        gpu = [ev[k].elapsed_time(ev[k + 1]) for k in range(20)]
        if el > 0.016:
            bad += 1
            hb = max(range(20), key=lambda k: sum(host[k]))
            gb = max(range(20), key=lambda k: gpu[k])
            print("round %3d: %.1f ms; worst host step %d copy/step/record = %.2f/%.2f/%.2f ms; worst gpu step %d = %.2f ms (median %.2f)"
                  % (rnd, el * 1e3, hb, host[hb][0] * 1e3, host[hb][1] * 1e3, host[hb][2] * 1e3, gb, gpu[gb], float(np.median(gpu))))
    print(mode, "stalled rounds:", bad, "of 150; median round %.2f ms" % (1e3 * float(np.median(tot))))
    sim.close()

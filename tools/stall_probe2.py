#!/usr/bin/env python3
"""developer tool: the bench's own sequence (create simulator, 5 warm-up steps, 20 timed steps), repeated, with host
time per call and a GPU event per step: where does the sporadic stall land?   gpurun -- python3 tools/stall_probe2.py [graph|events]"""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
stalls = 0
keep = []
for rep in range(reps):
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        sim = bench.make_sim(bench.scenes_for("waymo", 1024, 0), bench.params_for("waymo"), 64, 0)
        torch.cuda.synchronize()
        if "settle" in mode:
            time.sleep(0.3)
        batches = bench.action_batches(1024, 64, dev, seed=1)
        act = sim.action_tensor().to_torch()
        for k in range(5):
            act.copy_(batches[k]); sim.step()
        if "spin" in mode:   # keep the GPU busy for a while before the timed stretch
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.2:
                for k in range(8):
                    act.copy_(batches[k]); sim.step()
                torch.cuda.synchronize()
        if mode.startswith("events"):
            sim.kernel_timing(True)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        host = []
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(20):
            a = time.perf_counter()
            act.copy_(batches[k % 8])
            b = time.perf_counter()
            sim.step()
            c = time.perf_counter()
            ev[k + 1].record()
            host.append((a - t0, b - a, c - b, time.perf_counter() - c))
        t_sub = time.perf_counter() - t0
        t_poll = None
        if mode.endswith("poll"):
            while not ev[20].query():
                pass
            t_poll = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gpu = [ev[k].elapsed_time(ev[k + 1]) for k in range(20)]
        flag = el > 0.013
        stalls += flag
        flag = el * 1e3 > sum(gpu) + 2.0
        print("rep %2d %s: %.1f ms (host submit %.1f ms%s); gpu steps sum %.1f ms max %.2f" % (rep, mode, el * 1e3, t_sub * 1e3,
              "" if t_poll is None else ", last event seen by polling at %.1f ms" % (t_poll * 1e3), sum(gpu), max(gpu)),
              "STALL" if flag else "")
        if False:
            for k in range(20):
                if max(host[k][1:]) > 1e-3 or gpu[k] > 1.0:
                    print("    step %2d at +%.2f ms: host copy/step/record %.2f/%.2f/%.2f ms, gpu interval %.2f ms" %
                          (k, host[k][0] * 1e3, host[k][1] * 1e3, host[k][2] * 1e3, host[k][3] * 1e3, gpu[k]))
        if "keep" in mode:
            keep.append(sim)   # never freed: is the stall the driver still tearing down the previous simulator's memory?
        else:
            sim.close()
print(mode, "stalls:", stalls, "of", reps)

#!/usr/bin/env python3
"""developer tool: where the 20-step timed region's wall clock exceeds its kernels.  One simulator (bench scene, a reset of
all worlds every 91st step like bench.py), repeated 20-step stretches after different pauses; per stretch the wall clock, the
GPU time between per-step events, the largest interval and whether an episode reset fell inside (a second observation pass).
gpurun -- python3 tools/region_probe.py"""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for("synthetic", 1024, 0), bench.params_for("synthetic"), 64, 0)
    batches = bench.action_batches(1024, 64, dev, seed=1)
    worlds = list(range(1024))
    k = bench.run_steps(sim, batches, worlds, 40)
    torch.cuda.synchronize()
    for pause_ms in (0, 0, 0, 1, 1, 5, 5, 30, 30, 100, 100, 0, 0):
        k = bench.run_steps(sim, batches, worlds, 8, start=k)
        torch.cuda.synchronize()
        time.sleep(pause_ms / 1e3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        resets = sum(1 for q in range(k, k + 20) if (q + 1) % bench.EPISODE == 0)
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(20):
            k = bench.run_steps(sim, batches, worlds, 1, start=k)
            ev[_ + 1].record()
        t_sub = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = 1e3 * (time.perf_counter() - t0)
        gpu = [ev[q].elapsed_time(ev[q + 1]) for q in range(20)]
        print("pause %3d ms: wall %.2f ms (%.3f per step), host submit %.2f ms, event intervals sum %.2f first %.3f median %.3f max %.3f, "
              "episode resets inside: %d" % (pause_ms, el, el / 20, 1e3 * t_sub, sum(gpu), gpu[0], sorted(gpu)[10], max(gpu), resets))
    sim.close()

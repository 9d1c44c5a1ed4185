#!/usr/bin/env python3
"""developer tool: where the 20-step timed region's wall clock exceeds its kernels.  One simulator (bench scene), repeated
20-step stretches after different pauses; per stretch the wall clock, the GPU time between per-step events, and the largest
gap.   gpurun -- python3 tools/region_probe.py"""
import os, sys, time
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for("synthetic", 1024, 0), bench.params_for("synthetic"), 64, 0)
    batches = bench.action_batches(1024, 64, dev, seed=1)
    act = sim.action_tensor().to_torch()
    for k in range(40):
        act.copy_(batches[k % 8]); sim.step()
    torch.cuda.synchronize()
    for pause_ms in (0, 0, 0, 1, 1, 5, 5, 30, 30, 100, 100, 0, 0):
        for k in range(8):
            act.copy_(batches[k % 8]); sim.step()
        torch.cuda.synchronize()
        time.sleep(pause_ms / 1e3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(20):
            act.copy_(batches[k % 8]); sim.step()
            ev[k + 1].record()
        t_sub = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = 1e3 * (time.perf_counter() - t0)
        gpu = [ev[k].elapsed_time(ev[k + 1]) for k in range(20)]
        print("pause %3d ms: wall %.2f ms (%.3f per step), host submit %.2f ms, event intervals sum %.2f first %.3f median %.3f max %.3f" %
              (pause_ms, el, el / 20, 1e3 * t_sub, sum(gpu), gpu[0], sorted(gpu)[10], max(gpu)))
    sim.close()

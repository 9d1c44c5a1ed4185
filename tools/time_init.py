#!/usr/bin/env python3
"""developer tool: where SimManager construction spends its time (1024 worlds tiled from 8 synthetic scenes)."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
import torch
import madrona_gpudrive as mg
from gpudrive_lab_amd import synth
d = os.path.join(tempfile.gettempdir(), "gpudrive_amd_bench_scenes")
paths = synth.write_scenes(d, list(range(8)))
scenes = [paths[i % 8] for i in range(1024)]
p = mg.Parameters(); p.polylineReductionThreshold = 0.0; p.observationRadius = 50.0
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for rep in range(2):
    t = time.time(); sim = mg.SimManager(mg.madrona.ExecMode.CUDA, 0, scenes, p, max_agents=64); torch.cuda.synchronize()
    print("create %.3f s" % (time.time() - t))
    t = time.time(); sim.set_maps(scenes); torch.cuda.synchronize(); print("set_maps (same scenes) %.3f s" % (time.time() - t))
    t = time.time(); sim.reset(list(range(1024))); torch.cuda.synchronize(); print("reset all %.4f s" % (time.time() - t))
    sim.close()

#!/usr/bin/env python3
"""developer tool: time SimManager construction / set_maps over many DISTINCT scene files (copies of the
committed scenes), from JSON and from .gdsm caches."""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
import torch
import madrona_gpudrive as mg
from gpudrive_lab_amd import scene_cache

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
src = [os.path.join(ROOT, "tests", "data", f) for f in ("test.json", "tfrecord-00002-of-01000_407.json", "tfrecord-00000-of-01000_4.json")]
d = tempfile.mkdtemp()
paths = []
for i in range(n):
    p = os.path.join(d, "scene_%04d.json" % i)
    shutil.copy(src[i % 3], p)
    paths.append(p)
p = mg.Parameters(); p.polylineReductionThreshold = 0.1
t = time.time(); sim = mg.SimManager(mg.madrona.ExecMode.CUDA, 0, paths, p, max_agents=64); torch.cuda.synchronize(); t_json = time.time() - t
t = time.time(); sim.set_maps(paths[::-1]); torch.cuda.synchronize(); t_set = time.time() - t
t = time.time(); cached = scene_cache.build_cache(paths, 0.1); t_build = time.time() - t
t = time.time(); sim.set_maps(cached); torch.cuda.synchronize(); t_cache = time.time() - t
print("%d distinct scenes: create from JSON %.2f s, set_maps from JSON %.2f s, build caches %.2f s, set_maps from caches %.2f s"
      % (n, t_json, t_set, t_build, t_cache))
shutil.rmtree(d)

"""Seeded synthetic scenes in the reference's on-disk scene format (the JSON that
data_utils/process_waymo_files.py writes and src/json_serialization.hpp reads), for benchmark
worlds with an exact agent / road-segment count:

  * 64 vehicles on a jittered 8x8 grid, size U(4,5.5) x U(1.8,2.3), yaw U(-pi,pi), speed U(0,15),
    constant-velocity 91-step logged trajectories, all valid;
  * 4096 road-edge segments from 32 random-walk polylines (129 points, step U(0.5,5) m,
    turn N(0,0.1) rad per step).

With polylineReductionThreshold = 0 every segment becomes one road entity (R_w = 4096).
"""
import json
import math
import os

import numpy as np


def make_scene(seed, n_agents=64, n_polylines=32, pts_per_polyline=129):
    rng = np.random.default_rng(seed)
    side = int(math.ceil(math.sqrt(n_agents)))
    objects = []
    for i in range(n_agents):
        gx, gy = i % side, i // side
        x0 = (gx - side / 2) * 12.0 + rng.uniform(-3, 3) + 1000.0
        y0 = (gy - side / 2) * 12.0 + rng.uniform(-3, 3) - 500.0
        yaw = rng.uniform(-math.pi, math.pi)
        speed = rng.uniform(0.0, 15.0)
        vx, vy = speed * math.cos(yaw), speed * math.sin(yaw)
        pos = [{"x": x0 + vx * 0.1 * t, "y": y0 + vy * 0.1 * t, "z": 0.0} for t in range(91)]
        objects.append({
            "position": pos,
            "width": float(rng.uniform(1.8, 2.3)),
            "length": float(rng.uniform(4.0, 5.5)),
            "height": 1.6,
            "heading": [yaw] * 91,
            "velocity": [{"x": vx, "y": vy}] * 91,
            "valid": [True] * 91,
            "goalPosition": {"x": pos[-1]["x"], "y": pos[-1]["y"], "z": 0.0},
            "type": "vehicle",
            "id": i,
            "mark_as_expert": False,
        })
    roads = []
    for r in range(n_polylines):
        x = 1000.0 + rng.uniform(-80, 80)
        y = -500.0 + rng.uniform(-80, 80)
        th = rng.uniform(-math.pi, math.pi)
        geom = [{"x": x, "y": y, "z": 0.0}]
        for _ in range(pts_per_polyline - 1):
            th += rng.normal(0.0, 0.1)
            step = rng.uniform(0.5, 5.0)
            x += step * math.cos(th)
            y += step * math.sin(th)
            geom.append({"x": x, "y": y, "z": 0.0})
        roads.append({"geometry": geom, "type": "road_edge", "map_element_id": 15, "id": r})
    return {
        "name": "synthetic_%d.json" % seed,
        "scenario_id": "synth%011d" % seed,
        "objects": objects,
        "roads": roads,
        "tl_states": {},
        "metadata": {"sdc_track_index": 0, "objects_of_interest": [], "tracks_to_predict": []},
    }


def write_scenes(out_dir, seeds, **kw):
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for s in seeds:
        p = os.path.join(out_dir, "synthetic_%d.json" % s)
        if not os.path.exists(p):
            tmp = p + ".tmp%d" % os.getpid()
            with open(tmp, "w") as fh:
                json.dump(make_scene(s, **kw), fh)
            os.replace(tmp, p)
        paths.append(p)
    return paths

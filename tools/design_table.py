#!/usr/bin/env python3
"""developer tool: rewrite the numeric cells of DESIGN.md §6's results table from profiles/r03_bench_driver_command.json (first
figure of each cell) and profiles/r03_bench_default.json (in brackets); the other columns and the rest of the file stay."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(f):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    out = {}
    for r in [d] + d["other_workloads"]:
        out[r.get("workload", "synthetic")] = dict(ms=r["ms_per_step"], ev=r["ms_per_step_events"], ks=r["kernels_sum_us"] / 1e3,
                                                   rate=r.get("agent_steps_per_s", d["value"]) / 1e6,
                                                   road=r["kernels"]["k_map_obs+k_map_rows"]["avg_us"] / 1e3,
                                                   frac=100 * r["roofline"]["frac"], k=r["kernels"])
    return out


A = load(os.path.join(ROOT, "profiles", "r03_bench_driver_command.json"))
B = load(os.path.join(ROOT, "profiles", "r03_bench_default.json"))
ROWS = {"synthetic exact-64, R = 4096 (primary)": "synthetic", "Waymo tiles (35,489 live agents)": "waymo",
        "Waymo tiles, 4096 worlds, AgentStop + goal reward (config 3)": "cfg3", "Waymo tiles + 360° LiDAR (config 5)": "lidar",
        "Waymo tiles + BEV rasters": "bev", "synthetic, learner-side loop (`rl_loop`)": "rl_loop"}
SET = {"synthetic": "synthetic_set", "Waymo tiles": "waymo_set", "config 3": "cfg3_set"}
path = os.path.join(ROOT, "DESIGN.md")
lines = open(path).read().split("\n")
for n, line in enumerate(lines):
    if not line.startswith("| "):
        continue
    cells = line.split(" | ")
    if len(cells) < 9:
        continue
    name = cells[0][2:]
    key = ROWS.get(name) if "reference order" in cells[1] else SET.get(name) if "set order" in cells[1] else None
    if key is None:
        continue
    a, b = A[key], B[key]
    nd = 3 if key == "waymo_set" else 2
    fmt = lambda x, k=nd: ("%%.%df" % k) % x
    cells[2] = ("**%s** (%s)" if key == "synthetic" else "%s (%s)") % (fmt(a["ms"]), fmt(b["ms"]))
    cells[3] = "%s (%s)" % (fmt(a["ev"]), fmt(b["ev"]))
    cells[4] = fmt(a["ks"]) + (" + `k_pack_obs`, `k_episode_step`" if key == "rl_loop" else "")
    cells[5] = ("**%.1f M** (%.1f M)" if key == "synthetic" else "%.0f M (%.0f M)" if a["rate"] >= 100 else "%.1f M (%.1f M)") % (a["rate"], b["rate"])
    road = "%s ms" % fmt(a["road"], 3 if a["road"] < 0.1 else 2)
    if key == "lidar":
        road += " + `k_lidar` %.2f" % (a["k"]["k_lidar"]["avg_us"] / 1e3)
    if key == "bev":
        road += " + `k_bev` %.2f" % (a["k"]["k_bev"]["avg_us"] / 1e3)
    cells[6] = road
    cells[7] = "%.1f %%" % a["frac"]
    lines[n] = " | ".join(cells)
    print(lines[n][:200])
open(path, "w").write("\n".join(lines))

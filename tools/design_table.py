#!/usr/bin/env python3
"""developer tool: DESIGN.md section 6's results table (between the `<!-- results table -->` markers) from the committed bench
lines: profiles/r04_bench_driver_command.json (first figure of each cell), profiles/r04_bench_default.json (in brackets) and
profiles/r03_bench_driver_command.json (the round-3 column)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(f):
    d = json.loads(open(os.path.join(ROOT, "profiles", f)).read().strip().splitlines()[-1])
    out = {}
    for r in [dict(d, workload="synthetic", agent_steps_per_s=d["value"])] + d["other_workloads"]:
        out[r["workload"]] = r
    return out, d


A, lineA = load("r04_bench_driver_command.json")
B, _ = load("r04_bench_default.json")
C, _ = load("r03_bench_driver_command.json")
# not in round 3's line: timed this round on the build the round began with (gpurun_out of the first bench call)
EARLY = {"synthetic_128": "← 3.28 ms (this round's first build)", "waymo_raw": "← 3.65 ms (this round's first build)"}
ROWS = [("synthetic exact-64, R = 4096 (primary)", "synthetic", "reference order"),
        ("Waymo tiles (35,489 live agents, 6.5 K with roads in reach)", "waymo", "reference order"),
        ("Waymo tiles, 4096 worlds, AgentStop + goal reward (config 3)", "cfg3", "reference order"),
        ("Waymo tiles + 360° LiDAR (config 5)", "lidar", "reference order"),
        ("Waymo tiles + BEV rasters", "bev", "reference order"),
        ("synthetic, learner-side loop (`rl_loop`)", "rl_loop", "reference order"),
        ("synthetic, 128 agent slots (the fork's `kMaxAgentCount`), all live", "synthetic_128", "reference order"),
        ("Waymo tiles, unreduced polylines (5,191–10,000 roads per world)", "waymo_raw", "reference order"),
        ("synthetic", "synthetic_set", "set order"), ("Waymo tiles", "waymo_set", "set order"), ("config 3", "cfg3_set", "set order")]
out = ["| workload | mode | ms/step | events | Σ kernels | live agent-steps/s | road observation | roofline frac | round 3 |", "|---|---|---|---|---|---|---|---|---|"]
for name, key, mode in ROWS:
    a, b, c = A[key], B[key], C.get(key)
    nd = 3 if a["ms_per_step"] < 0.2 else 2
    f = lambda x: ("%%.%df" % nd) % x
    road = a["kernels"]["k_map_obs+k_map_rows"]["avg_us"] / 1e3
    extra = ""
    if key == "lidar":
        extra = " + `k_lidar` %.2f" % (a["kernels"]["k_lidar"]["avg_us"] / 1e3)
    if key == "bev":
        extra = " + `k_bev` %.2f" % (a["kernels"]["k_bev"]["avg_us"] / 1e3)
    rate = lambda r: "%.0f M" % (r["agent_steps_per_s"] / 1e6) if r["agent_steps_per_s"] >= 1e8 else "%.1f M" % (r["agent_steps_per_s"] / 1e6)
    prev = "← %s ms, %s, %.1f %%" % (f(c["ms_per_step"]), rate(c), 100 * c["roofline"]["frac"]) if c else EARLY.get(key, "")
    bold = "**%s**" if key == "synthetic" else "%s"
    out.append("| %s | %s | %s (%s) | %s (%s) | %s%s | %s (%s) | %s ms%s | %.1f %% | %s |" % (
        name, mode, bold % f(a["ms_per_step"]), f(b["ms_per_step"]), f(a["ms_per_step_events"]), f(b["ms_per_step_events"]),
        f(a["kernels_sum_us"] / 1e3), " + `k_pack_obs`, `k_episode_step`" if key == "rl_loop" else "", bold % rate(a), rate(b),
        ("%.3f" if road < 0.1 else "%.2f") % road, extra, 100 * a["roofline"]["frac"], prev))
cpu = lineA["cpu_baseline"]
out.append("| synthetic, CPU port, %d threads / all %d / 1 thread | — | — | — | — | %.3f M / %.3f M / %.3f M | — | — | |" % (
    cpu["cores"], cpu["host_cores"], cpu["value"] / 1e6, (cpu["all_cores_value"] or 0) / 1e6, cpu["single_thread_value"] / 1e6))
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
b0, b1 = "<!-- results table -->\n", "<!-- /results table -->\n"
i, j = s.index(b0) + len(b0), s.index(b1)
open(path, "w").write(s[:i] + "\n".join(out) + "\n" + s[j:])
print("\n".join(out))

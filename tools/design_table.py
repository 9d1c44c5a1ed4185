#!/usr/bin/env python3
"""developer tool: DESIGN.md section 6's results table (between the `<!-- results table -->` markers) from the committed bench
lines: profiles/r05_bench_driver_command.json and, for the last column, profiles/r04_bench_driver_command.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(f):
    d = json.loads(open(os.path.join(ROOT, "profiles", f)).read().strip().splitlines()[-1])
    out = {}
    for r in [dict(d, workload="synthetic", agent_steps_per_s=d["value"])] + d["other_workloads"]:
        out[r["workload"]] = r
    return out, d


A, lineA = load(sys.argv[1] if len(sys.argv) > 1 else "r05_bench_driver_command.json")
C, _ = load("r04_bench_driver_command.json")
ROWS = [("synthetic exact-64, R = 4096 (**primary**)", "synthetic", "k-NN, reference order"),
        ("same scenes", "synthetic_linear", "linear (callers' default)"),
        ("same scenes", "synthetic_set", "k-NN, set order"),
        ("same scenes, learner loop", "rl_loop", "reference order + pack written by the step"),
        ("same scenes, learner loop", "rl_loop_set", "set order + pack written by the step"),
        ("same generator, 128 slots, all live", "synthetic_128", "k-NN, reference order"),
        ("Waymo tiles (35,489 live agents)", "waymo", "k-NN, reference order"),
        ("Waymo tiles", "waymo_linear", "linear"),
        ("Waymo tiles", "waymo_set", "k-NN, set order"),
        ("`ppo_default`: Waymo tiles, 128 slots, `all_non_trivial`, vehicles only", "ppo_default", "linear"),
        ("config 3: 4096 worlds, AgentStop + goal reward", "cfg3", "k-NN, reference order"),
        ("config 3", "cfg3_set", "k-NN, set order"),
        ("Waymo tiles + 360° LiDAR (config 5)", "lidar", "reference order"),
        ("Waymo tiles + BEV rasters", "bev", "reference order"),
        ("Waymo tiles, unreduced polylines", "waymo_raw", "k-NN, reference order")]
out = ["| scenes | road selection | ms/step | live agent-steps/s | `k_world_step` µs | road observation µs | frac (bytes moved) | of the reference's bytes | agents left in place / step | round 4 ms/step |",
       "|---|---|---|---|---|---|---|---|---|---|"]
for name, key, mode in ROWS:
    if key not in A:
        continue
    a, c = A[key], C.get(key)
    rf = a["roofline"]
    road = [v["avg_us"] for k, v in a["kernels"].items() if k.startswith("k_map_obs")][0]
    extra = "".join(" + `%s` %.0f" % (k, a["kernels"][k]["avg_us"]) for k in ("k_lidar", "k_bev") if k in a["kernels"])
    rate = "%.0f M" % (a["agent_steps_per_s"] / 1e6) if a["agent_steps_per_s"] >= 1e8 else "%.1f M" % (a["agent_steps_per_s"] / 1e6)
    bold = "**%s**" if key == "synthetic" else "%s"
    out.append("| %s | %s | %s | %s | %.0f | %.0f%s | %.1f %% | %.0f %% | %.0f of %d | %s |" % (
        name, mode, bold % ("%.3f" % a["ms_per_step"]), bold % rate, a["kernels"]["k_world_step"]["avg_us"], road, extra, 100 * rf["frac"],
        100 * rf.get("frac_of_reference_bytes", rf["frac"]), rf.get("agents_skipped_per_launch", 0), a["live_agents_per_rank"],
        ("%.3f" % c["ms_per_step"]) if c else "— (new)"))
cpu = lineA.get("cpu_baseline")
if cpu:
    out.append("| synthetic, CPU port of the reference algorithm, %d threads / all %d / 1 thread | k-NN, reference order | — | %.3f M / %.3f M / %.3f M | | | | | | |" % (
        cpu["cores"], cpu["host_cores"], cpu["value"] / 1e6, (cpu["all_cores_value"] or 0) / 1e6, cpu["single_thread_value"] / 1e6))
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
b0, b1 = "<!-- results table -->\n", "<!-- /results table -->\n"
i, j = s.index(b0) + len(b0), s.index(b1)
open(path, "w").write(s[:i] + "\n".join(out) + "\n" + s[j:])
print("\n".join(out))

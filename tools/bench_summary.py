#!/usr/bin/env python3
"""developer tool: one line per workload of a bench.py JSON line.  tools/bench_summary.py gpurun_out/x.json [...]"""
import json, sys
for path in sys.argv[1:]:
    try:
        r = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:
        print(path, "unreadable:", e); continue
    print("==", path, "stamp", r.get("engine", {}).get("source_stamp"))
    rows = [dict(workload="(primary)", ms_per_step=r["ms_per_step"], agent_steps_per_s=r["value"], kernels=r["kernels"],
                 roofline=r["roofline"], ms_per_step_events=r.get("ms_per_step_events"))] + r.get("other_workloads", [])
    for o in rows:
        ks = " ".join("%s=%.1f" % (k.replace("k_map_obs+k_map_rows", "road").replace("k_world_step", "state"), v["avg_us"]) for k, v in o["kernels"].items())
        print("%-14s %.4f ms  %7.1f M/s  %s  frac %.4f" % (o["workload"], o["ms_per_step"], o["agent_steps_per_s"] / 1e6, ks, o["roofline"]["frac"]))

#!/usr/bin/env python3
"""developer tool: one line per workload of a bench.py JSON line.  tools/bench_rows.py gpurun_out/x.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    head = dict(workload=j["config"]["workload"][:24] + "..(primary)", ms_per_step=j["ms_per_step"], agent_steps_per_s=j["value"],
                roofline=j["roofline"], kernels=j["kernels"])
    print(f)
    for r in [head] + j["other_workloads"]:
        rf = r["roofline"]
        print("  %-28s %.3f ms/step %7.1f M a-s/s  road %6.1f us frac %.3f alg %5.0f MB skipped %6.0f ref %4.0f MB | %s"
              % (r["workload"], r["ms_per_step"], r["agent_steps_per_s"] / 1e6, rf["avg_kernel_us"], rf["frac"],
                 rf["algorithmic_bytes_per_launch"] / 1e6, rf.get("agents_skipped_per_launch", 0), rf.get("reference_bytes_per_launch", 0) / 1e6,
                 {k: round(v["avg_us"], 1) for k, v in r["kernels"].items()}))

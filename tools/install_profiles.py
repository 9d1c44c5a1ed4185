#!/usr/bin/env python3
"""Copy the summaries that tools/profile_all.sh left under gpurun_out/prof_<tag>/ into profiles/
(r01_<tag>_kernel_stats.csv, r01_<tag>_bench_under_rocprof.json, r01_<tag>_pmc_traffic.json) and
rebuild profiles/r01_traffic.json, the per-launch HBM bytes bench.py reports as roofline.traffic:
(2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 HBM note)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"
TAGS = ("exact_synthetic", "exact_waymo", "set_synthetic", "set_waymo", "lidar")
traffic = {}
for tag in TAGS:
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    if not os.path.isdir(src):
        print("missing", src)
        continue
    dst = os.path.join(ROOT, "profiles", "%s_%s_" % (ROUND, tag))
    shutil.copy(os.path.join(src, "kernel_stats_summary.csv"), dst + "kernel_stats.csv")
    with open(os.path.join(src, "bench_trace.json")) as fh:
        line = fh.read().strip().splitlines()[-1]
    with open(dst + "bench_under_rocprof.json", "w") as fh:
        fh.write(line + "\n")
    with open(os.path.join(src, "pmc_traffic_summary.json")) as fh:
        pmc = json.load(fh)
    with open(dst + "pmc_traffic.json", "w") as fh:
        json.dump(pmc, fh, indent=1)
    entry = {}
    for kern, c in pmc.items():
        f, w = c["FETCH_SIZE"]["mean"], c["WRITE_SIZE"]["mean"]
        entry[kern] = {"fetch_kib": f, "write_kib": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
    road = entry.get("k_map_obs", {})
    traffic[tag] = dict(road, kernels=entry)
with open(os.path.join(ROOT, "profiles", ROUND + "_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print(json.dumps({t: round(v.get("hbm_bytes_per_launch", 0) / 1e6, 1) for t, v in traffic.items()}))

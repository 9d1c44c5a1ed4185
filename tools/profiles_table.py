#!/usr/bin/env python3
"""Print the per-tag summary table of profiles/README.md from the installed <round>_* files."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
TAGS = ("exact_synthetic", "exact_waymo", "set_synthetic", "set_waymo", "lidar", "cfg3", "set_cfg3", "bev", "rl_loop",
        "exact_synthetic_128", "waymo_raw")
tr = json.load(open(os.path.join(ROOT, "profiles", R + "_traffic.json")))
print("| tag | road observation kernels (working launches, µs) | other | HBM bytes / step (PMC) | algorithmic bytes / step |")
print("|---|---|---|---|---|")
for t in TAGS:
    w = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_kernel_work.json" % (R, t))))
    b = json.loads(open(os.path.join(ROOT, "profiles", "%s_%s_bench_under_rocprof.json" % (R, t))).read())
    road = [(k, v) for k, v in w.items() if k.startswith(("k_knn_", "k_map_"))]
    other = [(k, v) for k, v in w.items() if k.startswith(("k_world_step<64, true", "k_world_step<128, true", "k_lidar", "k_bev", "k_pack", "k_episode"))]
    fmt = lambda kv: "%s %.0f%s" % (kv[0].split("<")[0], kv[1]["avg_us_working"],
                                    "" if kv[1]["working_launches"] >= 0.5 * kv[1]["launches"] else " (works in %d of %d launches)" % (kv[1]["working_launches"], kv[1]["launches"]))
    print("| `%s` | %s | %s | %.0f MB | %.0f MB |" % (t, " + ".join(fmt(x) for x in road), ", ".join(fmt(x) for x in other),
                                                    tr[t]["hbm_bytes_per_launch"] / 1e6, b["roofline"]["algorithmic_bytes_per_launch"] / 1e6))

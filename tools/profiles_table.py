#!/usr/bin/env python3
"""developer tool: the per-tag table of profiles/README.md's newest round from the installed summaries.
python tools/profiles_table.py [r05]   (prints the rows and replaces them in profiles/README.md between the table header of that
round and the paragraph after it)"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
TAGS = ["exact_synthetic", "synthetic_linear", "set_synthetic", "rl_loop_set", "rl_loop", "ppo_default", "exact_waymo", "waymo_linear",
        "set_waymo", "cfg3", "set_cfg3", "lidar", "bev", "exact_synthetic_128", "waymo_raw"]
tr = json.load(open(os.path.join(ROOT, "profiles", RND + "_traffic.json")))


def kn(n):
    m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", n)
    return m.group(1) if m else None


lines = []
for t in TAGS:
    f = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (RND, t))
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    work = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_kernel_work.json" % (RND, t))))
    road, other = [], []
    for r in rows:
        k = kn(r["Name"])
        if not k:
            continue
        us = work.get(k, {}).get("avg_us_working", float(r["AverageNs"]) / 1e3)
        if k.startswith(("k_map_obs", "k_map_rows", "k_knn")):
            if us >= 4.95 or k.startswith("k_map_obs_"):
                road.append("%s %.0f" % (k.split("<")[0] if not k.startswith("k_knn_rank") else k, us))
        elif k.startswith(("k_world_step<64, true", "k_world_step<128, true", "k_lidar", "k_bev", "k_pack_obs", "k_episode")):
            other.append("%s %.0f" % (k.split("<")[0], us))
    b = json.loads(open(os.path.join(ROOT, "profiles", "%s_%s_bench_under_rocprof.json" % (RND, t))).read().strip().splitlines()[-1])
    rf = b["roofline"]
    lines.append("| `%s` | %s | %s | %.0f MB | %.0f MB (%.0f MB counting every live agent) |" % (
        t, " + ".join(road), ", ".join(other), tr[t]["hbm_bytes_per_launch"] / 1e6, rf["algorithmic_bytes_per_launch"] / 1e6,
        rf.get("reference_bytes_per_launch", 0) / 1e6))
print("\n".join(lines))
p = os.path.join(ROOT, "profiles", "README.md")
s = open(p).read()
hdr = "| tag | road observation kernels (working launches, µs) | other | HBM bytes / step (PMC) | algorithmic bytes / step (agents left in place taken out) |\n|---|---|---|---|---|\n"
i = s.index(hdr) + len(hdr)
j = s.index("\n\n", i)
s = s[:i] + "\n".join(lines) + s[j:]
s = re.sub(r"All on source stamp `[0-9a-f]+`", "All on source stamp `%s`" % tr["source_stamp"], s)
open(p, "w").write(s)

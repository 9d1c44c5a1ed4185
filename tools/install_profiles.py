#!/usr/bin/env python3
"""Copy the summaries that tools/profile_all.sh left under gpurun_out/prof_<tag>/ into profiles/
(<round>_<tag>_kernel_stats.csv, _bench_under_rocprof.json, _pmc_traffic.json, _pmc_sq.json, _occupancy.json) and rebuild
profiles/<round>_traffic.json, the per-launch HBM bytes bench.py reports as roofline.traffic (means over the launches that did work: tools/profile.sh):
(2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 HBM note).  The road observation is
two launches (k_map_obs or k_map_obs_set, then k_map_rows); their bytes are summed.  The file is stamped with the source
hash of the build (bench.source_stamp): bench.py reports the traffic only for that build."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r05"
TAGS = ("exact_synthetic", "exact_waymo", "set_synthetic", "set_waymo", "lidar", "cfg3", "set_cfg3", "bev", "rl_loop",
        "exact_synthetic_128", "waymo_raw", "synthetic_linear", "waymo_linear", "ppo_default", "rl_loop_set")


def stamp():
    os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
    import bench
    return bench.source_stamp()


traffic = {"source_stamp": stamp()}
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
    for name in ("pmc_traffic_summary.json", "pmc_sq_summary.json", "kernel_work_summary.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), dst + name.replace("_summary", ""))
    # (registers / spills / waves per SIMD in it come from the compiler: tools/profile_summarize.py wrote it on the GPU box)
    shutil.copy(os.path.join(src, "occupancy.json"), dst + "occupancy.json")
    with open(os.path.join(src, "pmc_traffic_summary.json")) as fh:
        pmc = json.load(fh)
    # how often a kernel's launch does work at all (the trace run: k_map_obs is launched every step but selects only for the
    # groups the rank replay could not take -- in the steady state none; the gated reset pass of the learner-side loop)
    work = {}
    if os.path.exists(os.path.join(src, "kernel_work_summary.json")):
        with open(os.path.join(src, "kernel_work_summary.json")) as fh:
            wj = json.load(fh)
            # relative to the row kernel, which works in every pass that is not a gated empty one
            ref = max([v["working_launches"] for k, v in wj.items() if k.startswith(("k_map_rows", "k_map_obs_set"))] + [1])
            work = {k: min(1.0, v["working_launches"] / ref) for k, v in wj.items()}
    entry = {}
    for kern, c in pmc.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        f, w = c["FETCH_SIZE"]["mean"], c["WRITE_SIZE"]["mean"]
        share = work.get(kern, 1.0)
        entry[kern] = {"fetch_kib": f, "write_kib": w, "working_share_of_launches": share,
                       "hbm_bytes_per_launch": (2 * f + w) * 1024 * share}
    # the road observation: every kernel launch_map_obs issues (selection or scan / rank / replay / finish, fallback, rows)
    road = sum(v["hbm_bytes_per_launch"] for k, v in entry.items() if k.startswith(("k_map_obs", "k_map_rows", "k_knn_")))
    traffic[tag] = dict(hbm_bytes_per_launch=road, kernels=entry)
with open(os.path.join(ROOT, "profiles", ROUND + "_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print(json.dumps({t: round(v.get("hbm_bytes_per_launch", 0) / 1e6, 1) for t, v in traffic.items() if isinstance(v, dict)}))

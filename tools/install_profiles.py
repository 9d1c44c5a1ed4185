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
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"
TAGS = ("exact_synthetic", "exact_waymo", "set_synthetic", "set_waymo", "lidar", "cfg3", "set_cfg3", "bev", "rl_loop",
        "exact_synthetic_128", "waymo_raw")


def stamp():
    os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
    import bench
    return bench.source_stamp()


def occupancy_of(trace_dir):
    """Per kernel: registers, LDS and workgroup shape from the newest kernel trace, and the waves per SIMD they allow
    (VGPR granule 8, 512 per SIMD; LDS 160 KiB per CU shared by 4 SIMDs; hardware cap 8)."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    occ = {}
    if not files:
        return occ
    for row in csv.DictReader(open(files[-1])):
        m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", row.get("Kernel_Name", ""))
        if not m or m.group(1) in occ:
            continue
        g = lambda k: int(float(row.get(k, 0) or 0))
        vg, lds = g("VGPR_Count") + g("Accum_VGPR_Count"), g("LDS_Block_Size")
        wg = max(1, g("Workgroup_Size_X")) * max(1, g("Workgroup_Size_Y")) * max(1, g("Workgroup_Size_Z"))
        grid = max(1, g("Grid_Size_X")) * max(1, g("Grid_Size_Y")) * max(1, g("Grid_Size_Z"))
        waves_wg = (wg + 63) // 64
        alloc = -(-max(vg, 1) // 8) * 8
        by_vgpr = min(8, 512 // alloc)
        by_lds = min(8.0, (160 * 1024 // lds) * waves_wg / 4.0) if lds else 8.0
        occ[m.group(1)] = dict(vgpr=g("VGPR_Count"), agpr=g("Accum_VGPR_Count"), sgpr=g("SGPR_Count"), lds_bytes_per_workgroup=lds,
                               workgroup_size=wg, workgroups=grid // wg, waves_per_simd_by_vgpr=by_vgpr,
                               waves_per_simd_by_lds=by_lds, waves_per_simd=min(8.0, by_vgpr, by_lds))
    return occ


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
    with open(dst + "occupancy.json", "w") as fh:
        json.dump(occupancy_of(os.path.join(src, "trace")), fh, indent=1)
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

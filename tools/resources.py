#!/usr/bin/env python3
"""developer tool: registers, spills, LDS and occupancy of every kernel of one source file, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  tools/resources.py map_obs_rank.hip [extra flags]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(ROOT, "gpudrive_lab_amd", "csrc")


def analyse(src, extra=()):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                          "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + list(extra), cwd=d, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for ln in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", ln) or re.search(r"remark: +(.*?) \[-Rpass", ln)
        if not m:
            if "error" in ln: print(ln)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"gd::\(anonymous namespace\)::|\(gd::DevSim[^)]*\)|void ", "", cur)
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    return rows


if sys.argv[1] == "--json-all":
    # every kernel of the product library, with the source stamp bench.py prints: what tools/profile_summarize.py takes the
    # registers, spills and occupancy from (rocprofv3's VGPR_Count is HALF the allocation on gfx950)
    sys.path.insert(0, ROOT)
    import bench
    allk = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip"):
            for k, r in analyse(f).items():
                allk[k] = dict(source=f, vgpr=int(r.get("VGPRs", 0)), agpr=int(r.get("AGPRs", 0)), sgpr=int(r.get("TotalSGPRs", 0)),
                               vgpr_spills=int(r.get("VGPRs Spill", 0)), sgpr_spills=int(r.get("SGPRs Spill", 0)),
                               scratch_bytes_per_lane=int(r.get("ScratchSize [bytes/lane]", 0)),
                               lds_bytes_per_workgroup=int(r.get("LDS Size [bytes/block]", 0)),
                               waves_per_simd=int(r.get("Occupancy [waves/SIMD]", 0)))
    json.dump(dict(source_stamp=bench.source_stamp(), compiler="hipcc -O3 -Rpass-analysis=kernel-resource-usage, gfx950", kernels=allk),
              sys.stdout, indent=1)
    sys.exit(0)
src = sys.argv[1]
rows = analyse(src, sys.argv[2:])
print("%-44s %5s %5s %6s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "LDS", "occ"))
for k, r in rows.items():
    print("%-44s %5s %5s %6s %6s %7s %4s" % (k[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                              r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))

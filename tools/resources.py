#!/usr/bin/env python3
"""developer tool: registers, spills, LDS and occupancy of every kernel of one source file, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  tools/resources.py map_obs_rank.hip [extra flags]"""
import os, re, subprocess, sys
src = sys.argv[1]
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpudrive_lab_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:], cwd=d, capture_output=True, text=True).stderr
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
        cur = re.sub(r"gd::\(anonymous namespace\)::|\(gd::DevSim\)|void ", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print("%-44s %5s %5s %6s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "LDS", "occ"))
for k, r in rows.items():
    print("%-44s %5s %5s %6s %6s %7s %4s" % (k[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                              r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))

#!/usr/bin/env python3
"""developer tool: durations of consecutive launches of one kernel from a rocprofv3 kernel trace: gpurun -- python tools/kernel_durations.py <expt> <kernel substring> [workload]
(prints the mean duration of the 1st, 2nd, ... launch of that kernel within a step)"""
import csv, glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
expt, pat = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else "synthetic"
out = os.path.join(root, "gpurun_out", "kd_tmp")
subprocess.run(["rm", "-rf", out]); os.makedirs(out)
env = dict(os.environ, GPUDRIVE_DEV="1", GPUDRIVE_AMD_LIB=os.path.join(root, "build", "expt", "expt_%s.so" % expt), TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", "python3", os.path.join(root, "bench.py"), "--steps", "60", "--warmup", "10",
                "--no-cpu-baseline", "--workloads", wl], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=root)
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
seq, prev_match, k = {}, False, 0
for s, e, n in rows:
    if pat in n:
        k = k + 1 if prev_match else 0
        seq.setdefault(k, []).append(e - s)
        prev_match = True
    else:
        prev_match = False
for k, v in sorted(seq.items()):
    print("launch %d of a run of consecutive launches: %d launches, mean %.1f us" % (k + 1, len(v), sum(v) / len(v) / 1e3))

#!/bin/bash
# developer tool: who owns the host during the timed region of the driver's bench command?
#   gpurun -- tools/trace_gap.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gap
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workloads synthetic,waymo,cfg3,lidar > $OUT/plain.json 2>$OUT/plain.err
python3 - <<PY
import json
d=json.loads(open("$OUT/plain.json").read().strip().splitlines()[-1])
def show(w, ms, k): print("%-10s ms/step %.3f  kernels %s" % (w, ms, {n:round(v["avg_us"]) for n,v in k.items()}))
show("synthetic", d["ms_per_step"], d["kernels"])
for w in d["other_workloads"]: show(w["workload"], w["ms_per_step"], w["kernels"])
PY
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workloads waymo > $OUT/traced.json 2>$OUT/traced.err
python3 - <<PY
import csv,glob,collections
api=[]
for f in glob.glob("$OUT/trace/**/*hip_api_trace.csv", recursive=True):
    api+=list(csv.DictReader(open(f)))
ker=[]
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    ker+=list(csv.DictReader(open(f)))
print("api rows", len(api), "kernel rows", len(ker))
if api:
    print(list(api[0].keys()))
    for r in api: r["dur"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    # the last 20 launches of the state kernel bound the timed region
    ks=[r for r in ker if "k_world_step" in r["Kernel_Name"]]
    ks.sort(key=lambda r:int(r["Start_Timestamp"]))
    t_lo=int(ks[-20]["Start_Timestamp"])-5_000_000; t_hi=int(ks[-1]["End_Timestamp"])+1_000_000
    print("timed window ms", (t_hi-t_lo)/1e6)
    win=[r for r in api if int(r["End_Timestamp"])>=t_lo and int(r["Start_Timestamp"])<=t_hi]
    agg=collections.defaultdict(lambda:[0,0])
    for r in win:
        agg[r["Function"]][0]+=r["dur"]; agg[r["Function"]][1]+=1
    for fn,(d,n) in sorted(agg.items(), key=lambda x:-x[1][0])[:15]:
        print("%-40s calls %5d total %9.3f ms" % (fn,n,d/1e6))
    win.sort(key=lambda r:-r["dur"])
    for r in win[:25]:
        print("%-40s %9.3f ms at +%.3f ms" % (r["Function"], r["dur"]/1e6, (int(r["Start_Timestamp"])-t_lo)/1e6))
    kw=[r for r in ker if int(r["Start_Timestamp"])>=t_lo]
    kw.sort(key=lambda r:int(r["Start_Timestamp"]))
    prev=None
    for r in kw[:60]:
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        print("%-50s start +%9.3f dur %8.3f gap %8.3f" % (r["Kernel_Name"][:50], (s-t_lo)/1e6, (e-s)/1e6, (s-prev)/1e6 if prev else 0)); prev=e
PY

#!/bin/bash
# developer tool: does Python's collector own the sporadic host stall inside the 20-step timed region?
#   gpurun -- tools/gc_probe.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gc
for i in 1 2 3; do
  GPUDRIVE_BENCH_KEEP_GC=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workloads synthetic,waymo,cfg3,lidar,bev,rl_loop > gpurun_out/gc/keep_$i.json 2>gpurun_out/gc/keep_$i.err
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workloads synthetic,waymo,cfg3,lidar,bev,rl_loop > gpurun_out/gc/off_$i.json 2>gpurun_out/gc/off_$i.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/gc/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    rows=[("synthetic",d)]+[(w["workload"],w) for w in d["other_workloads"]]
    print(f.split("/")[-1], " ".join("%s %.2f/%.2f gc%.0f" % (n[:5], r["ms_per_step"], r["ms_per_step_events"], r["gc_ms_in_timed_stretches"]) for n,r in rows))
PY

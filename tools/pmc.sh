#!/bin/bash
# developer tool: PMC counters for the step kernels (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
shift
rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --roofline-steps 1 --no-cpu-baseline --workloads ${WORKLOAD:-synthetic} > $OUT/bench.log 2>$OUT/err.log
python3 - <<PY
import csv,glob,collections
files=glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "k_map_obs" in k or "k_world_step" in k:
            agg[k.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print("   %-28s n=%d mean=%.4g"%(c,len(vals),sum(vals)/len(vals)))
PY

#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then HBM
# traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, MI355X_MICROARCH.md "HBM").
# Run on the GPU box:  gpurun -- tools/profile.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 91 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2>$OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 4 --warmup 1 --roofline-steps 1 --no-cpu-baseline "$@" > $OUT/bench_pmc_$C.json 2>$OUT/pmc_$C.err
done
python3 - <<PY
import csv,glob,collections,json
out="$OUT"
stats=glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True)
rows=[]
for f in stats:
    rows+=list(csv.DictReader(open(f)))
with open(out+"/kernel_stats_summary.csv","w") as fh:
    if rows:
        w=csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:12]:
    print({k:(v[:70] if isinstance(v,str) else v) for k,v in r.items()})
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(out+"/pmc_%s/**/*counter_collection.csv"%C, recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]
            key=next((n for n in ("k_map_obs","k_world_step","k_lidar","k_bev") if n in k), None)
            if key: agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary={k:{c:{"mean":sum(x)/len(x),"n":len(x)} for c,x in v.items()} for k,v in agg.items()}
json.dump(summary, open(out+"/pmc_traffic_summary.json","w"), indent=1)
print(json.dumps(summary))
PY

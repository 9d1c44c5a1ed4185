#!/bin/bash
# developer tool: L1 / L2 request counters of the road-observation kernels (one rocprofv3 --pmc pass):  gpurun -- bash tools/pmc_l2.sh [bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
i=0
for G in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"; do
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 bench.py --steps 4 --warmup 2 --spin-ms 0 --no-align --no-cpu-baseline "$@" > $OUT/bench_g$i.log 2>$OUT/err_g$i.log
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z_]+(<[^>]*>)?)",row["Kernel_Name"])
        if m: agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name,v in agg.items():
    c={a:sum(x)/len(x) for a,x in v.items()}
    if c.get("TCC_REQ_sum",0) < 1e5: continue
    print("%-28s " % name + "  ".join("%s %.2fM" % (a.replace("_sum",""), b/1e6) for a,b in sorted(c.items())))
PY

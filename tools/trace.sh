#!/bin/bash
# developer tool: per-kernel statistics of one bench command:  gpurun -- tools/trace.sh <bench args>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_tmp
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 91 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench.json 2>$OUT/err.log
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print("%-70s calls %6s avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY

#!/bin/bash
# developer tool: time of k_knn_rank when it stops after phase n (GPUDRIVE_RANK_DBG; results wrong, timing only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# the phase switches exist only in a diagnostic build: tools/build_expt.sh diag -DGD_DIAG (before gpurun: the .so travels)
export GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$GRAFT_REPO_ROOT/build/expt/expt_diag.so
[ -f "$GPUDRIVE_AMD_LIB" ] || { echo "build it first: tools/build_expt.sh diag -DGD_DIAG"; exit 1; }
for D in ${PHASES:-1 2 3 4 5 0}; do
  OUT=gpurun_out/phase_$D; rm -rf $OUT; mkdir -p $OUT
  GPUDRIVE_RANK_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --workloads synthetic > $OUT/bench.json 2>$OUT/err.log
  python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))):
        if "k_knn_rank" in r["Name"] or "k_knn_replay" in r["Name"]: print("dbg $D %-50s avg %10.1f us" % (r["Name"][:50], float(r["AverageNs"])/1e3))
PY
done

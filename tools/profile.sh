#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ for ONE bench command (run on the GPU box):
#   gpurun -- tools/profile.sh <tag> [bench args...]
#   1. --kernel-trace --stats of the bench command        -> kernel_stats_summary.csv, occupancy.json (VGPR / LDS / waves)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (MI355X_MICROARCH.md "HBM")  -> pmc_traffic_summary.json
#   3. --pmc SQ counter groups, separate passes            -> pmc_sq_summary.json
# (no trace domains in the --pmc passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 91 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2>$OUT/trace.err
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 4 --warmup 1 --spin-ms 0 --no-align --no-cpu-baseline "$@" > $OUT/bench_pmc_$i.json 2>$OUT/pmc_$i.err
  i=$((i+1))
done
python3 tools/profile_summarize.py $OUT
# gpurun merges at most 64 MiB back: the raw traces and counter dumps (50-70 MB per tag) stay on the box, the summaries travel
rm -rf $OUT/trace $OUT/pmc_[0-9]*

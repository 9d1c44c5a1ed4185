#!/bin/bash
# developer tool: per-kernel averages of alternative builds, alternating:  gpurun -- tools/ab_trace.sh <kernel substring> <workload> <name> <name> ...
# (each name is traced in turn, the whole list twice, so that drift of the box shows as a difference between the two passes)
PAT=$1; WL=$2; shift 2
for pass in 1 2; do
  for e in "$@"; do
    bash tools/trace_expt.sh $e --workloads $WL > gpurun_out/ab_$e.txt 2>&1
    echo "$e pass $pass: $(grep "$PAT" gpurun_out/ab_$e.txt | tr -s ' ' | cut -c1-140 | tr '\n' ';')"
  done
done

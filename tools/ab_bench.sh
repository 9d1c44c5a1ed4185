#!/bin/bash
# developer tool: bench.py's HIP-event kernel averages for alternative builds (tools/build_expt.sh), alternating, the list twice:
#   gpurun -- tools/ab_bench.sh <workloads> <name|product>[:ENV=VALUE...] ...
WL=$1; shift
for pass in 1 2; do
  for spec in "$@"; do
    e=${spec%%:*}; envs=""
    [ "$spec" != "$e" ] && envs=$(echo "${spec#*:}" | tr ':' ' ')
    if [ "$e" = product ]; then pre=""; else pre="GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$PWD/build/expt/expt_$e.so"; fi
    env $pre $envs python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --workloads $WL > gpurun_out/abb_$e.json 2> gpurun_out/abb_$e.err
    echo "== $spec pass $pass"; python3 tools/bench_rows.py gpurun_out/abb_$e.json | tail -n +2
  done
done

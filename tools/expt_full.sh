#!/bin/bash
# developer tool: full default-length bench of alternative builds (build/expt/expt_<n>.so); "0" = the shipped library
for e in "$@"; do
  LIB=$GRAFT_REPO_ROOT/build/expt/expt_$e.so
  [ "$e" = "0" ] && LIB=$GRAFT_REPO_ROOT/gpudrive_lab_amd/libgpudrive_amd.so
  GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$LIB timeout -k 10 400 python bench.py --no-cpu-baseline --workloads ${WL:-synthetic,waymo} > gpurun_out/exptf_$e.log 2>gpurun_out/exptf_$e.err
  python -c "
import json;r=json.loads(open('gpurun_out/exptf_$e.log').read().strip().splitlines()[-1]);print('expt','$e','ms/step %.3f'%r['ms_per_step'],{k:round(v['avg_us']) for k,v in r['kernels'].items()},[(o['workload'],round(o['ms_per_step'],3),round(o['kernels']['k_map_obs+k_map_rows']['avg_us'])) for o in r['other_workloads']])"
done

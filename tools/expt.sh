#!/bin/bash
# developer tool: time alternative builds of the library (build/expt/expt_<n>.so, selected through GPUDRIVE_AMD_LIB)
for e in "$@"; do
  GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$GRAFT_REPO_ROOT/build/expt/expt_$e.so timeout -k 10 200 python bench.py --steps 91 --warmup 10 --roofline-steps 91 --no-cpu-baseline --workloads ${WL:-synthetic} > gpurun_out/expt_$e.log 2>gpurun_out/expt_$e.err
  python -c "
import json;r=json.loads(open('gpurun_out/expt_$e.log').read().strip().splitlines()[-1]);print('expt','$e','ms/step %.3f'%r['ms_per_step'],{k:round(v['avg_us']) for k,v in r['kernels'].items()})"
done

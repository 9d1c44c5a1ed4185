#!/bin/bash
# developer tool: time the road-observation kernel with parts switched off
for f in 0 1 2 3 4 7; do
  GPUDRIVE_DEBUG_FLAGS=$f timeout -k 10 200 python bench.py --steps 30 --warmup 5 --roofline-steps 20 --no-cpu-baseline --workloads ${1:-synthetic} > gpurun_out/abl_$f.log 2>/dev/null
  python -c "
import json;r=json.load(open('gpurun_out/abl_$f.log'));print('flags',$f,'ms/step %.3f'%r['ms_per_step'],{k:round(v['avg_us']) for k,v in r['kernels'].items()})"
done

#!/bin/bash
# developer tool: time the state kernel with parts switched off (8 = no partner rows, 16 = no collision)
for f in 0 8 16 24; do
  GPUDRIVE_DEBUG_FLAGS=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 --roofline-steps 20 --no-cpu-baseline --knn-order 1 --workloads ${1:-synthetic} > gpurun_out/abl_$f.log 2>/dev/null
  python -c "
import json;r=json.load(open('gpurun_out/abl_$f.log'));print('flags',$f,'ms/step %.3f'%r['ms_per_step'],{k:round(v['avg_us']) for k,v in r['kernels'].items()})"
done

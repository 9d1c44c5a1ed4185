#!/bin/bash
# developer tool: road-kernel time vs number of worlds (fills 1/8 .. 2 generations of the chip)
for W in 128 256 512 768 1024 2048; do
  timeout -k 10 200 python bench.py --worlds $W --steps 20 --warmup 3 --roofline-steps 10 --no-cpu-baseline --workloads ${1:-synthetic} > gpurun_out/occ_$W.log 2>gpurun_out/occ_$W.err
  python -c "
import json;r=json.loads(open('gpurun_out/occ_$W.log').read().strip().splitlines()[-1]);print('worlds',$W,'ms/step %.3f'%r['ms_per_step'],{k:round(v['avg_us']) for k,v in r['kernels'].items()})"
done

#!/bin/bash
GPUDRIVE_AMD_LIB=$GRAFT_REPO_ROOT/gpudrive_lab_amd/expt_9.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --roofline-steps 1 --no-cpu-baseline --workloads synthetic > gpurun_out/expt_9.log 2>gpurun_out/expt_9.err
grep "XW" gpurun_out/expt_9.log | head -10; grep XP gpurun_out/expt_9.log | tail -16
tail -1 gpurun_out/expt_9.log | python -c "
import json,sys;r=json.loads(sys.stdin.read());print({k:round(v['avg_us']) for k,v in r['kernels'].items()})"

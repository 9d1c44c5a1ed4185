#!/bin/bash
# developer tool: k_world_step with one phase skipped (GPUDRIVE_STEP_DBG; results wrong, timing only)
cd $GRAFT_REPO_ROOT
# the phase switches exist only in a diagnostic build: tools/build_expt.sh diag -DGD_DIAG (before gpurun: the .so travels)
export GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$GRAFT_REPO_ROOT/build/expt/expt_diag.so
[ -f "$GPUDRIVE_AMD_LIB" ] || { echo "build it first: tools/build_expt.sh diag -DGD_DIAG"; exit 1; }
for D in 0 1 2 3; do
  GPUDRIVE_STEP_DBG=$D python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --workloads ${WL:-synthetic_set,waymo_set} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dbg $D', ' | '.join('%s k_world_step %.1f us' % (r.get('workload','primary'), r['kernels']['k_world_step']['avg_us']) for r in [d]+d['other_workloads']))
"
done

#!/bin/bash
# developer tool: the set-order road observation under every schedule (rows fused or not x agents per wave):  gpurun -- bash tools/set_schedules.sh <expt build>
E=${1:-cur}
for WLN in synthetic_set waymo_set cfg3_set; do
  for F in 0 1; do for APW in 1 2 4 16; do
    R=$(GPUDRIVE_SET_FUSED_ROWS=$F GPUDRIVE_SET_AGENTS_PER_WAVE=$APW WL=$WLN bash tools/expt.sh $E | sed 's/^expt [a-z0-9]* //')
    echo "$WLN fused=$F apw=$APW $R"
  done; done
  R=$(WL=$WLN bash tools/expt.sh $E | sed 's/^expt [a-z0-9]* //'); echo "$WLN engine's choice: $R"
done

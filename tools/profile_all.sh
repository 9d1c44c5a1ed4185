#!/bin/bash
# Collect every profiles/ tag on the GPU box (about 16 minutes; the newest tags first):  gpurun --timeout 1200 -- tools/profile_all.sh [tags...]
TAGS=${@:-exact_synthetic synthetic_linear ppo_default set_synthetic rl_loop_set exact_waymo waymo_linear set_waymo cfg3 set_cfg3 lidar bev rl_loop exact_synthetic_128 waymo_raw}
for t in $TAGS; do
  case $t in
    exact_synthetic) a="--workloads synthetic";;
    exact_waymo) a="--workloads waymo";;
    set_synthetic) a="--workloads synthetic --knn-order 1";;
    set_waymo) a="--workloads waymo --knn-order 1";;
    lidar) a="--workloads lidar";;
    cfg3) a="--workloads cfg3";;
    set_cfg3) a="--workloads cfg3 --knn-order 1";;
    bev) a="--workloads bev";;
    rl_loop) a="--workloads rl_loop";;
    exact_synthetic_128) a="--workloads synthetic_128";;
    waymo_raw) a="--workloads waymo_raw";;
    synthetic_linear) a="--workloads synthetic_linear";;
    waymo_linear) a="--workloads waymo_linear";;
    ppo_default) a="--workloads ppo_default";;
    rl_loop_set) a="--workloads rl_loop_set";;
  esac
  timeout -k 10 400 bash tools/profile.sh $t $a > gpurun_out/prof_$t.log 2>&1 && echo "$t done" || echo "$t FAILED"
done

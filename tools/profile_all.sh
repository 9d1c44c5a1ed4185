#!/bin/bash
# Collect every profiles/ tag in one call on the GPU box:  gpurun --timeout 1200 -- tools/profile_all.sh
set -e
bash tools/profile.sh exact_synthetic --workloads synthetic > gpurun_out/prof_exact_synthetic.log 2>&1; echo "exact_synthetic done"
bash tools/profile.sh exact_waymo --workloads waymo > gpurun_out/prof_exact_waymo.log 2>&1; echo "exact_waymo done"
bash tools/profile.sh set_synthetic --workloads synthetic --knn-order 1 > gpurun_out/prof_set_synthetic.log 2>&1; echo "set_synthetic done"
bash tools/profile.sh set_waymo --workloads waymo --knn-order 1 > gpurun_out/prof_set_waymo.log 2>&1; echo "set_waymo done"
bash tools/profile.sh lidar --workloads lidar > gpurun_out/prof_lidar.log 2>&1; echo "lidar done"

#!/bin/bash
# developer tool: build tools/ubench/round_aw<N> (cycles per replace_top / per drain round); run the binaries on the GPU box
cd "$(dirname "$0")/../.."
for aw in ${@:-32}; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -DGD_MAP_OBS_AW=$aw -I gpudrive_lab_amd/csrc -o tools/ubench/round_aw$aw tools/ubench/round.hip gpudrive_lab_amd/csrc/kernels.hip gpudrive_lab_amd/csrc/bev_lidar.hip gpudrive_lab_amd/csrc/pack_obs.hip gpudrive_lab_amd/csrc/episode.hip gpudrive_lab_amd/csrc/engine.cpp gpudrive_lab_amd/csrc/scene.cpp gpudrive_lab_amd/csrc/scene_cache.cpp 2>&1 | grep -i "error"
done

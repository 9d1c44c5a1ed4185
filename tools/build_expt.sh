#!/bin/bash
# developer tool: build an alternative library build/expt/expt_<name>.so (outside the product package; _capi loads it only with GPUDRIVE_DEV=1) with extra compiler flags
# (e.g. tools/build_expt.sh t4 -DGD_TRIG_NUM=4; tools/build_expt.sh clk -DGD_CLOCKS for the phase clocks of k_knn_rank that
# EXPT=clk tools/rank_spikes.py prints; tools/build_expt.sh diag -DGD_DIAG for the phase switches
# GPUDRIVE_RANK_DBG / GPUDRIVE_STEP_DBG that tools/rank_phases.sh and tools/step_phases.sh use); time it on the GPU box with tools/expt.sh <name> ...
cd "$(dirname "$0")/../gpudrive_lab_amd/csrc" || exit 1
NAME=$1; shift
mkdir -p ../../build/expt
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" --offload-arch=gfx950 -shared -o ../../build/expt/expt_$NAME.so $(sed -n 's/^SRCS := //p' Makefile)

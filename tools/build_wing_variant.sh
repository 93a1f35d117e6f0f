#!/bin/bash
# usage: build_wing_variant.sh <name> [flags...]  -> tools/exp/libapg_wing_<name>.so
cd "$(dirname "$0")/.."
name=$1; shift; mkdir -p tools/exp
C=apg_trajectory_tracking_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DAPG_EXPERIMENT_BUILD "$@" -Iinclude -I$C -c $C/wing.hip -o tools/exp/wing_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "wing_rollout_lds_kernelILi0" | grep -E "VGPRs|AGPRs|Scratch|Occupancy" | tr '\n' ' '; echo
objs=$(ls $C/*.o | grep -v "/wing.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libapg_wing_$name.so $objs tools/exp/wing_$name.o

#!/bin/bash
# usage: build_gemm_variant.sh <name> [flags...]  -> tools/exp/libapg_gemm_<name>.so
cd "$(dirname "$0")/.."
name=$1; shift; mkdir -p tools/exp
C=apg_trajectory_tracking_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAPG_EXPERIMENT_BUILD "$@" -Iinclude -I$C -c $C/planes_gemm.hip -o tools/exp/planes_gemm_$name.o || exit 1
objs=$(ls $C/*.o | grep -v planes_gemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libapg_gemm_$name.so $objs tools/exp/planes_gemm_$name.o

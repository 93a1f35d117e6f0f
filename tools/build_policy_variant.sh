#!/bin/bash
# usage: build_policy_variant.sh <name> [flags...]  -> tools/exp/libapg_pol_<name>.so
# mlp_rollout.hip, mlp_concurrent.hip, lstm.hip and mlp_wing.hip (the kernels on policy_mfma.h) recompiled
# with extra flags, linked with the other shipped objects.
cd "$(dirname "$0")/.."
name=$1; shift; mkdir -p tools/exp
C=apg_trajectory_tracking_amd/csrc
for f in mlp_rollout mlp_concurrent lstm mlp_wing; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DAPG_EXPERIMENT_BUILD "$@" -Iinclude -I$C -c $C/$f.hip -o tools/exp/${f}_$name.o || exit 1
done
objs=$(ls $C/*.o | grep -v "/mlp_rollout.o\|/mlp_concurrent.o\|/lstm.o\|/mlp_wing.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libapg_pol_$name.so $objs tools/exp/mlp_rollout_$name.o tools/exp/mlp_concurrent_$name.o tools/exp/lstm_$name.o tools/exp/mlp_wing_$name.o

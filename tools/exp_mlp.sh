#!/bin/bash
# Build timing-experiment variants of mlp.hip (APG_MLP_EXP bit mask, see the
# top of mlp.hip) into tools/exp/libapg_exp<N>.so.  Run on the build box; the
# .so files travel with gpurun.  Select one with APG_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
python -m apg_trajectory_tracking_amd.build >/dev/null
mkdir -p tools/exp
C=apg_trajectory_tracking_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize \
    -DAPG_MLP_EXP=$n -I include -I $C -c $C/mlp.hip -o tools/exp/mlp_exp$n.o
  objs=$(ls $C/*.o | grep -v '/mlp.o')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libapg_exp$n.so $objs tools/exp/mlp_exp$n.o
done
ls -la tools/exp/*.so

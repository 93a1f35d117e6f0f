#!/bin/bash
# eager autoregressive step (ms) for the product library and every
# tools/exp/libapg_pol_ar*.so knock-out build (APG_AR_KNOCKOUT bits)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mode=${1:-autoregressive}
echo "product $(python tools/time_train_step.py $mode 2>/dev/null | tail -1)"
for f in tools/exp/libapg_pol_*.so; do
  echo "$(basename $f) $(APG_LIB=$PWD/$f python tools/time_train_step.py $mode 2>/dev/null | tail -1)"
done

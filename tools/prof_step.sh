#!/bin/bash
# rocprofv3 kernel statistics of one training step mode (eager launches):
#   tools/prof_step.sh autoregressive|LSTM|concurrent <out-prefix> [graph]
# writes gpurun_out/<out-prefix>_kernel_stats.csv and prints its top rows
mode=$1; out=$2; g=$3
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof -- python tools/time_train_step.py $mode $g > gpurun_out/_prof.log 2>&1
tail -1 gpurun_out/_prof.log
f=$(ls gpurun_out/_prof/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/${out}_kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${out}_kernel_stats.csv")))[:14]:
    print(r["Name"][:70].ljust(72), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
rm -rf gpurun_out/_prof gpurun_out/_prof.log

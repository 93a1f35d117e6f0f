#!/bin/bash
# rocprofv3 kernel statistics of any command:  tools/prof_cmd.sh <out-prefix> <command...>
out=$1; shift
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof -- "$@" > gpurun_out/_prof.log 2>&1
grep "ms/step" gpurun_out/_prof.log | tail -4
f=$(ls gpurun_out/_prof/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/${out}_kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${out}_kernel_stats.csv")))[:16]:
    print(r["Name"][:70].ljust(72), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
rm -rf gpurun_out/_prof gpurun_out/_prof.log

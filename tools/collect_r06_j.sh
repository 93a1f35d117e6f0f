#!/bin/bash
# round 6: per-kernel timeline of the fixed-wing training step
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/wstep
rm -rf gpurun_out/_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof -- python tools/time_wing_step.py > gpurun_out/wstep/log.txt 2>&1
grep "ms/step" gpurun_out/wstep/log.txt
t=$(ls gpurun_out/_prof/*/*kernel_trace.csv | head -1)
s=$(ls gpurun_out/_prof/*/*kernel_stats.csv | head -1)
cp "$s" gpurun_out/wstep/kernel_stats.csv
python tools/trace_step.py "$t" wing_policy_fwd > gpurun_out/wstep/timeline.txt
cat gpurun_out/wstep/timeline.txt
rm -rf gpurun_out/_prof

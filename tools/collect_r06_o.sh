# round 6, after collect_r06.sh: the parts that changed behind it (bench.py's secondary timings
# collect garbage before their runs; a stream-order step's _graphable() asks the launch form first;
# epoch timelines show a batch from the middle of an epoch)
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
rm -f $O/run_epoch.jsonl
for m in concurrent autoregressive LSTM; do
  timeout 300 python tools/time_run_epoch.py $m graph 32 >> $O/run_epoch.jsonl 2>/dev/null
  timeout 300 python tools/time_run_epoch.py $m eager 32 >> $O/run_epoch.jsonl 2>/dev/null
  k=mlp_concurrent_fwd_kernel; [ $m = autoregressive ] && k=mlp_rollout_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 8 > /dev/null 2>&1
  python tools/trace_step.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k 4 > $O/run_epoch_${m}_timeline.txt; rm -rf $O/re
done
cat $O/run_epoch.jsonl | cut -c1-120,200-330
timeout 600 python -m pytest tests -m gpu -x -q -k "epoch or graph or launch or wing" 2>&1 | tail -2

# round 6: one batch from the END of a 32-batch epoch per mode (clear of the next epoch's permutation sort)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
for m in concurrent autoregressive LSTM; do
  k=mlp_concurrent_fwd_kernel; [ $m = autoregressive ] && k=mlp_rollout_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 32 > /dev/null 2>&1
  python tools/trace_step.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k 2 > $O/run_epoch_${m}_timeline.txt
  [ $m != autoregressive ] && python tools/trace_epoch.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k 32 > $O/epoch_${m}.txt
  rm -rf $O/re; echo == $m; cat $O/run_epoch_${m}_timeline.txt
done

# round 6: the rows-path kernels with shuffled rows and with consecutive rows (what the row
# gather costs in DRAM line efficiency vs what its place in the kernel's prologue costs)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06n; mkdir -p $O
for m in concurrent LSTM; do
  k=lstm_rollout_fwd_kernel; [ $m = concurrent ] && k=mlp_concurrent_fwd_kernel
  for ns in "" 1; do
  rm -rf $O/re; APG_EPOCH_NO_SHUFFLE=$ns rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 32 noprefetch > /dev/null 2>&1
  echo "== $m no_shuffle=$ns"; python tools/trace_epoch.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k 32 | grep "us/batch" | head -4
  rm -rf $O/re
  done
done 2>&1 | tee $O/rows_noshuffle.txt

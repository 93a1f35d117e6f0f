# round 6: the LSTM step with the gate weight-gradient kernel (conv columns recomputed)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
for form in eager graph; do python tools/time_train_step.py LSTM $form 2>/dev/null | tail -1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py LSTM > $O/train_step_LSTM.txt 2>/dev/null
python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) lstm_rollout_fwd_kernel > $O/step_LSTM_timeline.txt; cat $O/step_LSTM_timeline.txt; cp $(ls $O/cs/*/*kernel_stats.csv | head -1) $O/step_LSTM_kernel_stats.csv; rm -rf $O/cs
python tools/time_run_epoch.py LSTM graph 32 2>/dev/null | tail -2

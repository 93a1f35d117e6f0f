# round 6: the LSTM step with the gate weight-gradient kernel (conv inputs recomputed)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
for form in eager graph; do python tools/time_train_step.py LSTM $form 2>/dev/null | tail -1; done | tee $O/train_step_LSTM.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py LSTM > /dev/null 2>&1
python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) lstm_rollout_fwd_kernel > $O/step_LSTM_timeline.txt; cat $O/step_LSTM_timeline.txt; cp $(ls $O/cs/*/*kernel_stats.csv | head -1) $O/step_LSTM_kernel_stats.csv; rm -rf $O/cs
for pf in noprefetch prefetch; do python tools/time_run_epoch.py LSTM graph 32 $pf 2>/dev/null | tail -1; done | tee $O/run_epoch_LSTM.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ce -- python tools/time_run_epoch.py LSTM graph 8 noprefetch > /dev/null 2>&1
python tools/trace_step.py $(ls $O/ce/*/*kernel_trace.csv | head -1) lstm_rollout_fwd_kernel > $O/run_epoch_LSTM_timeline.txt; cat $O/run_epoch_LSTM_timeline.txt; rm -rf $O/ce
bash tools/ab_gate_wgrad.sh product 2>&1 | grep "lstm_" | tee $O/gate_wgrad_kernel.txt
bash tools/pmc_step.sh LSTM $O/pmc > /dev/null 2>&1; cp $O/pmc/report.txt $O/pmc_lstm_step.txt; rm -rf $O/pmc; grep -A3 "^lstm_gate" $O/pmc_lstm_step.txt | cut -c1-200

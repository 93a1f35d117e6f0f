# round 6: the LSTM step with one sum launch behind both weight-gradient products and the tail's pack out of LDS
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "lstm or LSTM or recurrent" 2>&1 | tail -2
timeout 200 python tools/time_train_step.py LSTM graph 2>/dev/null | tail -1
rm -rf $O/cs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py LSTM > /dev/null 2>&1
python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) lstm_rollout_fwd_kernel | tee $O/step_LSTM_timeline.txt; rm -rf $O/cs

# round 6: the LSTM step's tail - parity and timing
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -k "lstm or invalidate" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log; tail -15 $O/pytest_new.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_trainers.py tests/test_gpu_fullsize.py tests/test_gpu_round5.py tests/test_gpu_in_sweep_recurrent.py -x -q -k "lstm or LSTM or launch_form or recurrent" > $O/pytest_old.log 2>&1; echo "rc=$?" >> $O/pytest_old.log; tail -5 $O/pytest_old.log | cut -c1-300
for form in eager graph; do python tools/time_train_step.py LSTM $form 2>/dev/null | tail -1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py LSTM > $O/train_step_LSTM.txt 2>/dev/null
python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) lstm_rollout_fwd_kernel > $O/step_LSTM_timeline.txt; cat $O/step_LSTM_timeline.txt; rm -rf $O/cs

# round 6: LSTM tail (second build), wing clock attribution, full GPU suite
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -k "lstm or invalidate" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log; tail -4 $O/pytest_new.log | cut -c1-300
for form in eager graph; do python tools/time_train_step.py LSTM $form 2>/dev/null | tail -1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py LSTM > $O/train_step_LSTM.txt 2>/dev/null
python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) lstm_rollout_fwd_kernel > $O/step_LSTM_timeline.txt; cat $O/step_LSTM_timeline.txt; rm -rf $O/cs
APG_LIB=$PWD/tools/exp/libapg_wing_clock.so timeout 300 python tools/wing_clock.py > $O/wing_clock.jsonl 2>$O/wing_clock.err; cat $O/wing_clock.jsonl | cut -c1-900
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log | cut -c1-300

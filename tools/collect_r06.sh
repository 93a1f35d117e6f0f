# The one script behind the final profiles/r06_* (run on the GPU box through gpurun;
# every result lands in gpurun_out/r06/, what is judged is copied to profiles/).
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
# 1. the bench line of this commit
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
# 2. headline-only kernel stats (the roofline kernel's average duration)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl -- python bench.py --headline-only --steps 20 --warmup 5 --min-ms 40 > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/hl/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/hl
# 3. the stream floors of the headline launch: bench.py's probe alone, and rounds 2-3's C++ harness
python tools/stream_floor.py > $O/stream_floor.json 2>/dev/null
tools/exp/hbm_probe calib > $O/hbm_probe_calib.txt 2>&1; cat $O/hbm_probe_calib.txt
# 4. the three fused steps: per-kernel stats and one step's timeline
for m in concurrent autoregressive LSTM; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py $m graph > $O/train_step_$m.txt 2>/dev/null
  k=mlp_concurrent_fwd_kernel; [ $m = autoregressive ] && k=mlp_rollout_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) $k > $O/step_${m}_timeline.txt
  cp $(ls $O/cs/*/*kernel_stats.csv | head -1) $O/step_${m}_kernel_stats.csv; rm -rf $O/cs
done
# 5. run_epoch: ms per batch and one batch's timeline in the default launch form
for m in concurrent autoregressive LSTM; do
  python tools/time_run_epoch.py $m graph 32 >> $O/run_epoch.jsonl 2>/dev/null
  python tools/time_run_epoch.py $m eager 32 >> $O/run_epoch.jsonl 2>/dev/null
  k=mlp_concurrent_fwd_kernel; [ $m = autoregressive ] && k=mlp_rollout_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 8 > /dev/null 2>&1
  python tools/trace_step.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k 4 > $O/run_epoch_${m}_timeline.txt; rm -rf $O/re
done
# 5b. whole epochs: per-batch kernel sums, idle gaps, what stands between two epochs
for m in concurrent LSTM; do
  k=mlp_concurrent_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 32 > /dev/null 2>&1
  python tools/trace_epoch.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k 32 > $O/epoch_${m}.txt; rm -rf $O/re
done
# 6. PMC of the three steps at this commit (counters only)
bash tools/pmc_step.sh concurrent $O/pmc_conc > $O/pmc_conc.log 2>&1; cp $O/pmc_conc/report.txt $O/pmc_concurrent_step.txt
bash tools/pmc_step.sh autoregressive $O/pmc_ar > $O/pmc_ar.log 2>&1; cp $O/pmc_ar/report.txt $O/pmc_ar_step.txt
bash tools/pmc_step.sh LSTM $O/pmc_lstm > $O/pmc_lstm.log 2>&1; cp $O/pmc_lstm/report.txt $O/pmc_lstm_step.txt
rm -rf $O/pmc_conc $O/pmc_ar $O/pmc_lstm
# 7. the GPU suite with the arbiter's statistics
python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "rc_all=$?" >> $O/pytest_all.log
grep "fp64 arbiter\|row arbiter" $O/pytest_all.log > $O/arbiter.txt
tail -3 $O/pytest_all.log
ls -la $O

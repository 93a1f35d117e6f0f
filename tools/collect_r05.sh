# The one script behind profiles/r05_* (run on the GPU box through gpurun; every
# result lands in gpurun_out/r05/, what is judged is copied to profiles/).
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; rm -rf $O; mkdir -p $O
# 1. the bench line of this commit
python bench.py > $O/bench.json 2> $O/bench.err
# 2. headline-only kernel stats (the roofline kernel's average duration)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl -- python bench.py --headline-only --steps 20 --warmup 5 --min-ms 40 > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/hl/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/hl
# 3. the three fused steps: per-kernel stats and one step's timeline
for m in concurrent autoregressive LSTM; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py $m graph > $O/train_step_$m.txt 2>/dev/null
  k=mlp_concurrent_fwd_kernel; [ $m = autoregressive ] && k=mlp_rollout_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) $k > $O/step_${m}_timeline.txt
  cp $(ls $O/cs/*/*kernel_stats.csv | head -1) $O/step_${m}_kernel_stats.csv; rm -rf $O/cs
done
# 4. run_epoch: ms per batch and one batch's timeline in the default launch form
for m in concurrent autoregressive LSTM; do
  python tools/time_run_epoch.py $m graph 32 >> $O/run_epoch.jsonl 2>/dev/null
  python tools/time_run_epoch.py $m eager 32 >> $O/run_epoch.jsonl 2>/dev/null
  k=mlp_concurrent_fwd_kernel; [ $m = autoregressive ] && k=mlp_rollout_fwd_kernel; [ $m = LSTM ] && k=lstm_rollout_fwd_kernel
  rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 8 > /dev/null 2>&1
  python tools/trace_step.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k > $O/run_epoch_${m}_timeline.txt; rm -rf $O/re
done
# 5. rows against gather, in-sweep against planes (A/B in one process each)
python tools/ab_rows.py both 300 > $O/ab_rows.txt 2>/dev/null
for w in "ar in" "ar planes" "lstm planes"; do python tools/ab_in_sweep.py $w >> $O/ab_in_sweep.txt 2>/dev/null; done
# 6. PMC of the concurrent step (four passes, counters only)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- python tools/time_train_step.py concurrent graph > $O/p$i.log 2>&1
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f step 2>/dev/null | grep -i "concurrent\|kernel\|wgrad" | cut -d, -f2,4,5 | head -12 >> $O/pmc_concurrent_step.txt
  rm -rf $O/p$i $O/p$i.log
done
# 7. the parameter-gradient row arbiter's statistics and the GPU suite
python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "rc_all=$?" >> $O/pytest_all.log
grep "fp64 arbiter\|operand range\|row arbiter\|rows:" $O/pytest_all.log > $O/arbiter.txt
tail -3 $O/pytest_all.log
ls -la $O

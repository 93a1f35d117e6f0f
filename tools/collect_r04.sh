set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04z; mkdir -p $O
# 1. full GPU suite with the arbiter statistics
python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "rc_all=$?" >> $O/pytest_all.log
grep "fp64 arbiter\|operand range" $O/pytest_all.log > $O/arbiter.txt
# 2. the bench line
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# 3. headline-only kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl -- python bench.py --headline-only --steps 20 --warmup 5 --min-ms 40 > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/hl/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/hl
# 4. concurrent step timeline (graph replay, in-sweep) + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs -- python tools/time_train_step.py concurrent graph > $O/train_step_concurrent.txt 2>/dev/null
python tools/trace_step.py $(ls $O/cs/*/*kernel_trace.csv | head -1) mlp_concurrent_fwd_kernel > $O/concurrent_step_timeline.txt
cp $(ls $O/cs/*/*kernel_stats.csv | head -1) $O/concurrent_step_kernel_stats.csv; rm -rf $O/cs
# 5. run_epoch: the default (epoch graph where it applies), per-step plans / graphs, eager; per-batch kernel list
for m in concurrent autoregressive LSTM; do
  python tools/time_run_epoch.py $m graph 32 prefetch epoch >> $O/run_epoch_final.jsonl 2>/dev/null
  python tools/time_run_epoch.py $m graph 32 noprefetch noepoch >> $O/run_epoch_final.jsonl 2>/dev/null
  python tools/time_run_epoch.py $m eager 32 noprefetch noepoch >> $O/run_epoch_final.jsonl 2>/dev/null
done
for m in concurrent LSTM autoregressive; do k=to_soa_kernel; [ $m = concurrent ] && k=mlp_concurrent_fwd_kernel; rocprofv3 --kernel-trace --output-format csv -d $O/re -- python tools/time_run_epoch.py $m graph 8 > /dev/null 2>&1; python tools/trace_step.py $(ls $O/re/*/*kernel_trace.csv | head -1) $k > $O/run_epoch_${m}_timeline.txt; rm -rf $O/re; done
python tools/ab_graph_alternation.py > $O/ab_graph_alternation.jsonl 2>/dev/null
python tools/ab_weight_products.py time > $O/ab_weight_products.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wp -- python tools/ab_weight_products.py time > /dev/null 2>&1
grep -i "concurrent\|wgrad\|pack_step" $(ls $O/wp/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 >> $O/ab_weight_products.txt; rm -rf $O/wp
# 6. packed step timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pk -- python bench.py --steps 5 --warmup 2 --min-ms 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
cp $(ls $O/pk/*/*kernel_stats.csv | head -1) $O/bench_full_kernel_stats.csv; rm -rf $O/pk
# 7. in-sweep A/B numbers
python tools/ab_concurrent_step.py > $O/ab_concurrent_step.jsonl 2>/dev/null
ls -la $O

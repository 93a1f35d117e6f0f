set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06i; rm -rf $O; mkdir -p $O
# concurrent reverse kernel: identity transposition (product) against the round-5 kernel
for rep in 1 2; do
for f in product tools/exp/libapg_pol_swapped.so; do
  n=$(basename $f .so)
  if [ $f = product ]; then unset APG_LIB; else export APG_LIB=$PWD/$f; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ko -- python tools/time_train_step.py concurrent graph > $O/conc_$n.txt 2>/dev/null
  s=$(ls $O/ko/*/*kernel_stats.csv | head -1)
  echo "$n $(grep mlp_concurrent_bwd_tm_kernel $s | cut -d, -f2-4) $(tail -1 $O/conc_$n.txt)" >> $O/ab_conc.txt
  rm -rf $O/ko
done; done
unset APG_LIB
cat $O/ab_conc.txt
# LSTM: the step before round 6 against the tail launch, same box
for rep in 1 2; do
for form in eager graph; do
  echo "legacy $(APG_STEP_LEGACY=1 python tools/time_train_step.py LSTM $form 2>/dev/null | tail -1)" >> $O/ab_lstm.txt
  echo "tail   $(python tools/time_train_step.py LSTM $form 2>/dev/null | tail -1)" >> $O/ab_lstm.txt
done; done
cat $O/ab_lstm.txt
timeout 1200 python -m pytest tests/test_gpu_in_sweep.py "tests/test_gpu_round5.py::test_parameter_gradient_rows_vs_fp64_at_full_size" "tests/test_gpu_fullsize.py::test_quad_concurrent_fused_full_size_vs_fp64_oracle" tests/test_gpu_trainers.py -x -q -k "concurrent or in_sweep or rows or G3 or two_sgd" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log | cut -c1-300

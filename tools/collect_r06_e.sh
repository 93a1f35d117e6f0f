# round 6: same-box A/B of the AR reverse sweep variants (kernel us under rocprof, step ms)
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
for f in product tools/exp/libapg_pol_*.so product; do
  n=$(basename $f .so)
  if [ $f = product ]; then unset APG_LIB; else export APG_LIB=$PWD/$f; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ko -- python tools/ab_in_sweep.py ar in > $O/ko_$n.txt 2>/dev/null
  s=$(ls $O/ko/*/*kernel_stats.csv | head -1)
  echo "$n $(grep mlp_rollout_bwd_tm_kernel $s | cut -d, -f2-4) $(tail -1 $O/ko_$n.txt)" >> $O/ab_raw.txt
  rm -rf $O/ko
done
unset APG_LIB
cat $O/ab_raw.txt
# parity of the ring variants (the product library is the identity build)
for f in tools/exp/libapg_pol_ringbc.so; do
APG_LIB=$PWD/$f timeout 900 python -m pytest tests/test_gpu_in_sweep_recurrent.py "tests/test_gpu_round5.py::test_parameter_gradient_rows_vs_fp64_at_full_size" "tests/test_gpu_fullsize.py::test_recurrent_fused_full_size_vs_fp64_oracle" -x -q > $O/pytest_$(basename $f .so).log 2>&1; echo "rc=$?" >> $O/pytest_$(basename $f .so).log
tail -3 $O/pytest_$(basename $f .so).log
done

# round 6: parity + timing of a kernel change in the autoregressive reverse sweep
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_in_sweep_recurrent.py tests/test_gpu_in_sweep.py "tests/test_gpu_round5.py::test_parameter_gradient_rows_vs_fp64_at_full_size" "tests/test_gpu_fullsize.py::test_recurrent_fused_full_size_vs_fp64_oracle" "tests/test_gpu_fullsize.py::test_quad_concurrent_fused_full_size_vs_fp64_oracle" tests/test_gpu_round6.py -x -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep "passed\|failed\|rc=\|Error" $O/pytest.log | cut -c1-600
grep "row arbiter" $O/pytest.log > $O/arbiter.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ko -- python tools/ab_in_sweep.py ar in > $O/ab_ar.txt 2>/dev/null
s=$(ls $O/ko/*/*kernel_stats.csv | head -1); grep "mlp_rollout" $s | cut -d, -f1-4; cat $O/ab_ar.txt; rm -rf $O/ko
for m in concurrent autoregressive; do python tools/time_train_step.py $m graph 2>/dev/null | tail -1; done

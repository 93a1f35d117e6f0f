# round 6, second GPU call: transposition probe, AR knock-outs incl. the load bits,
# the new parity tests
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
timeout 300 tools/exp/transposition_probe > $O/transposition_probe.jsonl 2> $O/transposition_probe.err; cat $O/transposition_probe.jsonl
for f in product tools/exp/libapg_pol_arko*.so; do
  n=$(basename $f .so)
  if [ $f = product ]; then unset APG_LIB; else export APG_LIB=$PWD/$f; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ko -- python tools/ab_in_sweep.py ar in > $O/ko_$n.txt 2>/dev/null
  s=$(ls $O/ko/*/*kernel_stats.csv | head -1)
  echo "$n $(grep mlp_rollout_bwd_tm_kernel $s | cut -d, -f1-4) $(grep 'mlp_rollout_fwd_kernel' $s | cut -d, -f1-4)" >> $O/ar_knockouts_raw.txt
  rm -rf $O/ko
done
unset APG_LIB
cat $O/ar_knockouts_raw.txt
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q -s > $O/pytest_r6.log 2>&1; echo "rc=$?" >> $O/pytest_r6.log
grep "row arbiter\|passed\|failed\|rc=" $O/pytest_r6.log | cut -c1-3000

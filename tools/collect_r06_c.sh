# round 6, third GPU call: transposition probe (fixed-point accumulators), L2 warm-up
# variants of the AR reverse sweep, stream floor from graph replays
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
timeout 300 tools/exp/transposition_probe > $O/transposition_probe.jsonl 2> $O/transposition_probe.err; cat $O/transposition_probe.jsonl
timeout 300 python tools/stream_floor.py > $O/stream_floor.json 2> $O/stream_floor.err; head -12 $O/stream_floor.json
for f in product tools/exp/libapg_pol_artouch*.so product; do
  n=$(basename $f .so)
  if [ $f = product ]; then unset APG_LIB; else export APG_LIB=$PWD/$f; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ko -- python tools/ab_in_sweep.py ar in > $O/ko_$n.txt 2>/dev/null
  s=$(ls $O/ko/*/*kernel_stats.csv | head -1)
  echo "$n $(grep mlp_rollout_bwd_tm_kernel $s | cut -d, -f2-4) $(tail -1 $O/ko_$n.txt)" >> $O/ar_touch_raw.txt
  rm -rf $O/ko
done
unset APG_LIB
cat $O/ar_touch_raw.txt

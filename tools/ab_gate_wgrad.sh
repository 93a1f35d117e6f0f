# round 6: the LSTM gate weight-gradient kernel, product build and knock-out variants
#   bash tools/ab_gate_wgrad.sh <variant> ...   (tools/exp/libapg_pol_<variant>.so; "product" = the shipped library)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  rm -rf gpurun_out/_abg
  if [ "$v" = product ]; then unset APG_LIB; else export APG_LIB=$PWD/tools/exp/libapg_pol_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_abg -- python tools/time_train_step.py LSTM > gpurun_out/_abg.log 2>&1
  python - "$v" <<PY
import csv, glob, sys
f = glob.glob("gpurun_out/_abg/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("lstm_gate_wgrad_kernel", "lstm_rollout_fwd", "lstm_rollout_bwd")):
        print(sys.argv[1].ljust(10), r["Name"][:40].ljust(42), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done
rm -rf gpurun_out/_abg gpurun_out/_abg.log

# kernel averages (rocprofv3 --stats) of one training step for variant builds
#   bash tools/ab_step_kernels.sh <mode> <variant> ...   ("product" = the shipped library)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mode=$1; shift
for v in "$@"; do
  rm -rf gpurun_out/_abs
  if [ "$v" = product ]; then unset APG_LIB; else export APG_LIB=$PWD/tools/exp/libapg_pol_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_abs -- python tools/time_train_step.py $mode > gpurun_out/_abs.log 2>&1
  python - "$v" <<PY
import csv, glob, sys
f = glob.glob("gpurun_out/_abs/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(sys.argv[1].ljust(10), r["Name"].replace("apg::(anonymous namespace)::", "")[:44].ljust(46), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done
rm -rf gpurun_out/_abs gpurun_out/_abs.log

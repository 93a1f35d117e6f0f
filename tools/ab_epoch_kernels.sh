# kernel averages of one concurrent epoch batch (rows path) for variant builds
#   bash tools/ab_epoch_kernels.sh <mode> <variant> ...   ("product" = the shipped library)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mode=$1; shift
k=lstm_rollout_fwd_kernel; [ $mode = concurrent ] && k=mlp_concurrent_fwd_kernel
for v in "$@"; do
  if [ "$v" = product ]; then unset APG_LIB; else export APG_LIB=$PWD/tools/exp/libapg_pol_$v.so; fi
  for ns in "" 1; do
    rm -rf gpurun_out/_abe
    APG_EPOCH_NO_SHUFFLE=$ns rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_abe -- python tools/time_run_epoch.py $mode graph 32 noprefetch > /dev/null 2>&1
    echo "$v shuffle=$([ -z "$ns" ] && echo yes || echo no) $(python tools/trace_epoch.py $(ls gpurun_out/_abe/*/*kernel_trace.csv | head -1) $k 32 | grep "us/batch" | head -2 | awk '{printf "%s %s  ", $1, $5}')"
  done
done
rm -rf gpurun_out/_abe

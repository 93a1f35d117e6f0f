# round 6 probe: rows of the next batch touched ahead (cache warm-up) - does the forward kernel's gather get cheaper?
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06n; mkdir -p $O
for how in before beside none; do
  timeout 120 python tools/touch_probe.py $how 2>&1 | tail -1
  rm -rf $O/tp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp -- timeout 120 python tools/touch_probe.py $how > /dev/null 2>&1
  python - $how <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r06n/tp/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("   ", sys.argv[1].ljust(8), r["Name"].replace("apg::(anonymous namespace)::", "")[:44].ljust(46), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
  rm -rf $O/tp
done 2>&1 | tee $O/touch_probe.txt

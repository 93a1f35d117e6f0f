cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk -- python tools/ab_weight_products.py time > gpurun_out/abk.log 2>&1
tail -4 gpurun_out/abk.log | cut -c1-80
python - <<EOF
import csv,glob
f=glob.glob("gpurun_out/abk/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("concurrent","wgrad","pack_step")): print(n[32:70].ljust(40), r["Calls"], round(float(r["AverageNs"])/1e3,1))
EOF
rm -rf gpurun_out/abk

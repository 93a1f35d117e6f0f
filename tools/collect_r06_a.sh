# round 6, first GPU call: suite, bench line, plain-command --gpus 2 on a 1-GPU box,
# PMC of the autoregressive step
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2.out 2>&1; echo "gpus2 rc=$?" >> $O/bench_gpus2.out; tail -2 $O/bench_gpus2.out
python bench.py --gpus 2 --steps 20 --warmup 5 --dry-run-cpu > $O/bench_gpus2_dry.out 2>&1; echo "gpus2 dry rc=$?" >> $O/bench_gpus2_dry.out
bash tools/pmc_step.sh autoregressive $O/pmc_ar > $O/pmc_ar.log 2>&1
cp $O/pmc_ar/report.txt $O/pmc_ar_step.txt
tail -c 1500 $O/bench.json

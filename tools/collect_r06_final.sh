# round 6, last call: the suite, smoke(), and bench.py as ONE launched rank on a live
# RCCL group (the multi-GPU code path with world size 1)
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06z; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --train-steps 80 --no-cpu-baseline > $O/bench_launched_world1.json 2> $O/bench_launched_world1.err; echo "launched rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06z/bench_launched_world1.json") if l.startswith("{")][-1])
print({k: (d[k].get("ms_per_step"), d[k].get("launch")) if isinstance(d.get(k), dict) else d.get(k) for k in ("train_step","train_step_ar","train_step_lstm")})
print(d["steps_summary"])
PY

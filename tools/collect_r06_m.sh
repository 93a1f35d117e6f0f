# round 6, after the LSTM weight-gradient kernel: bench.py's default run + the rocprof summary of the same command
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06m; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06m/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("stream_floor_us"))
print(d["steps_summary"])
PY

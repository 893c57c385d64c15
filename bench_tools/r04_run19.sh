# round 4, GPU call 19: the whole GPU suite and the default line (four verified sub-records) of the committed code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run19; rm -rf $E; mkdir -p $E
( time timeout 1700 python -m pytest tests -m gpu -q --durations=15 ) > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $E/bench_default.json 2> $E/bench_default.err
tail -30 $E/pytest_gpu.log
python - <<PY
import json
d = json.loads([l for l in open("$E/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("driver_run_s"))
for k, v in d.get("sub_records", {}).items():
    print(k, v.get("ms_per_step"), v.get("value"), (v.get("config") or {}).get("verified"), v.get("wall_s"), v.get("error"))
PY
tail -4 $E/bench_default.err

# round 4, GPU call 18: the folded key's context kept by its parent - f3 tests (incl. two proofs under one key), the compressing proof
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run18; rm -rf $E; mkdir -p $E
timeout 1500 python -m pytest tests/test_gpu_spartan.py tests/test_gpu_ipa.py -q -x -k "not key_fold" > $E/pytest_f3.log 2>&1; echo "pytest rc $?" >> $E/pytest_f3.log
tail -4 $E/pytest_f3.log
{
  python bench.py --workload compress --steps 5 --warmup 2 --no-cpu-baseline --verify
  python bench.py --workload compress --steps 5 --warmup 2 --no-cpu-baseline
} > $E/compress.jsonl 2> $E/compress.err
python - <<PY
import json
for l in open("$E/compress.jsonl"):
    d = json.loads(l)
    print(d["ms_per_step"], d["config"].get("verified"))
PY
tail -3 $E/compress.err
LURK_PROF_TIMELINE=$E/tl_compress.txt python bench.py --workload compress --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
L=[x.split() for x in open("$E/tl_compress.txt") if not x.startswith("#")]
t_end=float(L[-1][1]); rows=[x for x in L if float(x[0])>t_end-75000]
rows.sort(key=lambda x: float(x[0]))
t0=float(rows[0][0]); prev_end=None; prev_name=None
for x in rows:
    a=float(x[0])-t0; b=float(x[1])-t0
    if prev_end is not None and a-prev_end>250: print("gap %6.0f us  after %-20s (ended %8.0f) before %s" % (a-prev_end, prev_name, prev_end, x[4]))
    prev_end=max(b, prev_end or 0); prev_name=x[4]
PY

# round 4, GPU call 14: the key folded after four inner-product rounds - parity tests, the compressing proof with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run14; rm -rf $E; mkdir -p $E
timeout 1500 python -m pytest tests/test_gpu_ipa.py -q -x > $E/pytest_ipa.log 2>&1; echo "pytest rc $?" >> $E/pytest_ipa.log
tail -5 $E/pytest_ipa.log
{
  python bench.py --workload compress --steps 5 --warmup 2 --no-cpu-baseline --verify
  LURK_IPA_FOLD_MIN_LOG=0 python bench.py --workload compress --steps 5 --warmup 2 --no-cpu-baseline
  python bench.py --workload compress --steps 5 --warmup 2 --no-cpu-baseline
} > $E/compress.jsonl 2> $E/compress.err
python - <<PY
import json
for l in open("$E/compress.jsonl"):
    d = json.loads(l)
    print(d["ms_per_step"], d["config"].get("verified"), d.get("phase_ms") or d["config"].get("phase_ms") or {k: v for k, v in d.items() if "ms" in k})
PY
tail -3 $E/compress.err

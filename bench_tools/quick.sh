cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_msm.py -x -q 2>&1 | tail -2
B="--no-cpu-baseline --pmc off --no-plain-leg"
for a in "--steps 10 --warmup 3 --pipeline 1" "--steps 20 --warmup 5 --pipeline 3" "--steps 20 --warmup 5 --pipeline 3" "--steps 40 --warmup 5 --pipeline 3"; do
python bench.py $a $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('kernel_ms_per_commit_sync'))"
done

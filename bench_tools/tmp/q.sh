cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_msm.py tests/test_gpu_ipa.py -x -q 2>&1 | tail -2
B="--no-cpu-baseline --pmc off --no-plain-leg"
for a in "--log-n 13 --precompute 0" "--log-n 13 --precompute 1" "--log-n 20 --precompute 0" "--log-n 22 --precompute 0" "--log-n 22 --precompute 1"; do
python bench.py --steps 20 --warmup 5 --pipeline 1 $a $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d.get('kernel_ms_per_commit_sync'), d.get('sync_ms_per_commit'))"
done

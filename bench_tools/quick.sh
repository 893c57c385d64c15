cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_ipa.py tests/test_gpu_spartan.py -x -q 2>&1 | tail -3
for a in "--ipa-resident-key 1" "--ipa-resident-key 0" "--ipa-resident-key 1 --precompute 0"; do
python bench.py --workload compress --steps 3 --warmup 1 --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['kernels_ms_per_proof'])"
done

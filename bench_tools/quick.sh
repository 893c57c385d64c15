cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d.get('oneshot_host_pointers'), indent=1))"

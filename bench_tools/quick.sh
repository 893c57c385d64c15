cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --workload fold_step --steps 12 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E $*', d['value'], d['ms_per_step'], d['host_ms_per_step'])"; }
E=tfirst; run; run; run --rc 900 --steps 5 --warmup 2
export LURK_STEP_W_FIRST=1; E=wfirst; run; run; run --rc 900 --steps 5 --warmup 2
unset LURK_STEP_W_FIRST
python -m pytest tests/test_gpu_step.py -x -q 2>&1 | tail -2

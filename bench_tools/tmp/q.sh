cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --workload fold_step --steps 12 --warmup 3 --no-cpu-baseline --secondary 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E $*', d['value'], d['ms_per_step'], d['host_ms_per_step'])"; }
for rep in 1 2; do
export LURK_STEP_ORDER=0; E=order0; run
export LURK_STEP_ORDER=1; E=order1; run
done
export LURK_STEP_ORDER=0; E=order0; run --rc 900 --steps 5 --warmup 2
export LURK_STEP_ORDER=1; E=order1; run --rc 900 --steps 5 --warmup 2

cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_step.py tests/test_gpu_msm.py -x -q 2>&1 | tail -2
run() { python bench.py --workload fold_step --steps 12 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E $*', d['value'], d['ms_per_step'], d['host_ms_per_step'])"; }
E=fg; run; run
run --stage-ahead 1 --late-ranges 0
run --stage-ahead 1 --late-ranges 1
run --witness-ahead 0
run --rc 900 --steps 5 --warmup 2

cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_step.py -x -q 2>&1 | tail -1
for i in 1 2; do
python bench.py --workload fold_step --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d.get('secondary_curve_step',{}).get('ms_per_step'), d.get('both_curves_ms_per_step'), d.get('both_curves_iterations_per_s'))"
done

cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="--no-cpu-baseline --pmc off --no-plain-leg"
run() { python bench.py --steps 20 --warmup 5 --pipeline 3 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
E=default142; run
export LURK_MSM_ACC_WAVES=2 LURK_MSM_MAX_ACC=1; E=w2_max1; run
export LURK_MSM_ACC_WAVES=2 LURK_MSM_MAX_ACC=2; E=w2_max2; run
unset LURK_MSM_ACC_WAVES LURK_MSM_MAX_ACC
done

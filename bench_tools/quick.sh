cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="--no-cpu-baseline --pmc off --no-plain-leg"
for rep in 1 2 3 4 5 6 7 8; do
for a in "--steps 20 --warmup 5 --pipeline 2" "--steps 20 --warmup 5 --pipeline 3"; do
python bench.py $a $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done

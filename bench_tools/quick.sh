cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="--no-cpu-baseline --pmc off --no-plain-leg"
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --pipeline 2" "--steps 10 --warmup 3 --pipeline 1" "--steps 20 --warmup 5 --log-n 20" "--steps 20 --warmup 5 --log-n 20 --pipeline 2"; do
python bench.py $a $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done

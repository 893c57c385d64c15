#!/bin/bash
# Round measurement sweep (through gpurun): one JSON line per configuration into gpurun_out/sweep_<tag>.jsonl
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sweep_$TAG.jsonl; : > $OUT
run() { python bench.py "$@" 2>/dev/null | tail -1 >> $OUT; }
run --steps 20 --warmup 5                                         # headline as the driver runs it: 2^22 table, 3 in flight (+ live PMC traffic, plain leg, one-shot leg, cpu baseline)
run --steps 40 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg
run --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --pipeline 2
run --steps 10 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --pipeline 1
run --steps 10 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --pipeline 1 --precompute 0
run --steps 10 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --dist witness
run --steps 20 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --log-n 20 --pipeline 2
run --steps 20 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --log-n 20 --pipeline 1
run --steps 20 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --log-n 20 --pipeline 1 --precompute 0
run --steps 20 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --log-n 20 --pipeline 1 --precompute 0 --dist witness
run --steps 10 --warmup 2 --no-cpu-baseline --pmc off --no-plain-leg --log-n 22 --verify
run --gpus 2 --backend gloo --verify --log-n 20 --steps 5 --warmup 1 --no-cpu-baseline   # two ranks sharing this box's one GPU: functional line only
run --workload fold_step --steps 10 --warmup 2
run --workload fold_step --steps 10 --warmup 2 --no-cpu-baseline --stage-ahead 1 --late-ranges 0
run --workload fold_step --steps 10 --warmup 2 --no-cpu-baseline --stage-ahead 1 --late-ranges 1
run --workload fold_step --steps 5 --warmup 2 --rc 900 --no-cpu-baseline
run --workload poseidon_tree --steps 3 --warmup 1
run --workload ntt --log-n 24 --steps 10 --warmup 2
run --workload ntt --log-n 20 --steps 20 --warmup 2 --no-cpu-baseline
run --workload compress --steps 3 --warmup 1 --no-cpu-baseline
run --workload compress --steps 2 --warmup 1 --no-cpu-baseline --ipa-resident-key 0
python - <<PY
import json
for l in open("$OUT"):
    try: d=json.loads(l)
    except Exception as e: print("bad line", l[:100]); continue
    print(d["config"].get("workload","")[:70], "|", d["config"].get("commitments_in_flight", d["config"].get("staged_ahead", d.get("ipa", ""))), "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms", "| verified" if d.get("verified") else "")
PY

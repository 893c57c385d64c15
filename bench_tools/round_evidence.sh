# One GPU call that regenerates the round's measurement evidence (run through gpurun; results land in gpurun_out/evidence_<tag>/ and,
# for the summaries that are tracked, in gpurun_out/evidence_<tag>/profiles/ - copy those into profiles/ and commit them).
# usage: bash bench_tools/round_evidence.sh <tag> [notests]        (tag e.g. r06)
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/evidence_$TAG
rm -rf $E; mkdir -p $E/profiles
# 0. the whole GPU suite and smoke()
if [ "${2:-}" != "notests" ]; then
  ( time timeout 1300 python -m pytest tests -m gpu -q ) > $E/profiles/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/profiles/${TAG}_pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1; echo "smoke rc $?" >> $E/smoke.log
  tail -4 $E/profiles/${TAG}_pytest_gpu.log; tail -2 $E/smoke.log
fi
# 1. the default line exactly as the driver runs it
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $E/bench_default.json 2> $E/bench_default.err; echo "bench rc $?" >> $E/bench_default.err
grep '^{' $E/bench_default.json | tail -1 > $E/profiles/${TAG}_default_bench_line.json
# 2. rocprofv3 kernel trace + stats + PMC passes of the same workload, condensed (launch classes, traffic three ways)
bash bench_tools/profile.sh ${TAG}_msm_n22 > $E/profile_n22.log 2>&1
python bench_tools/summarize_profile.py gpurun_out/prof_${TAG}_msm_n22 ${TAG}_msm_n22_table 22 > $E/summarize_n22.log 2>&1
cp profiles/${TAG}_msm_n22_table* $E/profiles/ 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_msm_n22
# 3. the sweep: every workload / operating point DESIGN.md section 5 quotes, one JSON line each
Q="--no-cpu-baseline --pmc off --no-plain-leg --sub-records off"
{
  for cfg in "20 1" "20 2" "22 1" "22 3" "22 4"; do set -- $cfg; python bench.py --log-n $1 --pipeline $2 --steps 20 --warmup 5 $Q; done
  LURK_MSM_REDUCE_WAVE=0 python bench.py --log-n 20 --pipeline 1 --steps 20 --warmup 5 $Q   # one launch per reduction level (rounds 3-5)
  LURK_MSM_REDUCE_WAVE=0 python bench.py --log-n 20 --pipeline 2 --steps 20 --warmup 5 $Q
  python bench.py --log-n 22 --dist witness --steps 20 --warmup 5 $Q
  python bench.py --log-n 22 --precompute 0 --pipeline 1 --steps 10 --warmup 3 $Q
  python bench.py --gpus 2 --backend gloo --scaling strong --log-n 22 --steps 10 --warmup 3 $Q --verify   # two ranks on this box's one GPU: the strong-scaling path, functional
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --verify
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --stage-ahead 0 --no-cpu-baseline                      # W2 committed inside its own step (rounds 1-5)
  LURK_FOLD_CACHED_PRODUCTS=0 python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --stage-ahead 0 --no-cpu-baseline   # ... and the six-gather cross term: round 5's step
  LURK_FOLD_STAGED_MODE=2 python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline              # staged flow, BACKGROUND class instead of FOLLOW
  for q in 4 6 12; do GPU_MAX_HW_QUEUES=$q python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline; done   # hardware queues per priority class (default of the workload: 8)
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --devices 0,0 --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --devices 0,0 --auto-slices 1 --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 900 --steps 8 --warmup 2 --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 900 --steps 8 --warmup 2 --no-cpu-baseline --secondary 0 --stage-ahead 0
  python bench.py --workload store_hydrate --steps 10 --warmup 2 --verify
  python bench.py --workload compress --steps 10 --warmup 3 --verify
  python bench.py --workload compress --steps 10 --warmup 3 --spartan-prover python --no-cpu-baseline
  LURK_MSM_REDUCE_WAVE=0 python bench.py --workload compress --steps 10 --warmup 3 --no-cpu-baseline
  python bench.py --workload poseidon_tree --steps 5 --warmup 2 --no-cpu-baseline
  python bench.py --workload ntt --log-n 24 --steps 10 --warmup 3 --no-cpu-baseline
  python bench.py --workload ntt --log-n 20 --steps 10 --warmup 3 --no-cpu-baseline
} > $E/profiles/${TAG}_sweep.jsonl 2> $E/sweep.err
# 3b. rocprofv3 kernel stats of the other workloads (one short run each), and SQ_INSTS_VALU of the Poseidon tree for its issue budget
for wl in "fold_step --rc 100 --secondary 0" "compress" "poseidon_tree --log-n 24" "ntt --log-n 24"; do
  name=$(echo $wl | cut -d' ' -f1)
  ( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$name -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $E/profiles/${TAG}_${name}_bench_under_rocprof.json 2> $E/prof_$name.err )
  cp $(ls $E/prof_$name/*/*_kernel_stats.csv | head -1) $E/profiles/${TAG}_${name}_kernel_stats.csv 2>/dev/null; rm -rf $E/prof_$name
done
( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $E/pmc_tree -- python bench.py --workload poseidon_tree --log-n 24 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $E/pmc_tree.err )
python - <<PY > $E/profiles/${TAG}_poseidon_tree_sq_insts.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob("$E/pmc_tree/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "poseidon" not in k: continue
        i = 0 if r["Counter_Name"] == "SQ_INSTS_VALU" else 1 if r["Counter_Name"] == "SQ_WAVES" else None
        if i is None: continue
        acc[k][i] += float(r["Counter_Value"]); acc[k][2] += (i == 0)
print("# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU over bench.py --workload poseidon_tree --log-n 24 --steps 2 --warmup 1 (3 trees of 2 396 745 hash8)")
for k, (v, w, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(k, "launches", n, "SQ_INSTS_VALU %.1f M" % (v / 1e6), "SQ_WAVES %.0f" % w, "VALU wave-instructions per wave %.0f" % (v / max(w, 1)))
PY
rm -rf $E/pmc_tree
# 3c. the fold kernels in isolation (HIP events per launch) and the device-side scope timeline of the default step
{ python bench_tools/fold_bench.py 100; python bench_tools/fold_bench.py 900; } > $E/profiles/${TAG}_fold_kernels_isolated.txt 2> $E/fold_bench.err
LURK_PROF_TIMELINE=$E/step_scopes.txt python bench.py --workload fold_step --steps 12 --warmup 4 --no-cpu-baseline --secondary 0 > /dev/null 2> $E/step_tl.err
{ echo "# device-side scope timeline (LURK_PROF_TIMELINE, bench_tools/scope_timeline.py) of one step of bench.py --workload fold_step (rc = 100, default flow: --stage-ahead 3,"; echo "# GPU_MAX_HW_QUEUES=8, primary curve only); microseconds relative to the step's cross term: start, end, duration, stream, scope"; python bench_tools/scope_timeline.py $E/step_scopes.txt; } > $E/profiles/${TAG}_step_timeline_rc100.txt 2>&1
# 3d. cold starts of the staged flow at rc = 900, every output against the oracle after two steps (the flow whose first late-range commitment
# raced a NULL-stream memset until round 6)
{ echo "# 12 processes of: bench.py --workload fold_step --rc 900 --secondary 0 --no-cpu-baseline --verify --steps 2 --warmup 0 (exit code, ms per step)"
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    python bench.py --workload fold_step --rc 900 --secondary 0 --no-cpu-baseline --verify --steps 2 --warmup 0 > $E/cold.json 2> $E/cold.err; rc=$?
    python -c "
import json,sys
try:
    d=json.loads([l for l in open('$E/cold.json') if l.startswith('{')][-1]); print('run $i rc $rc', d['ms_per_step'], 'verified', (d['config'].get('verified') or {}).get('ok'))
except Exception as e: print('run $i rc $rc no line', e)"
  done; } > $E/profiles/${TAG}_rc900_cold_starts.txt 2>&1
./bench_tools/microbench > $E/profiles/${TAG}_microbench_instr_rates.txt 2>&1
./bench_tools/mds_mfma > $E/profiles/${TAG}_mds_mfma_final.txt 2>&1
tail -c 400 $E/bench_default.json; echo; tail -3 $E/bench_default.err; wc -l $E/profiles/${TAG}_sweep.jsonl
python - <<PY
import json
for l in open("$E/profiles/${TAG}_sweep.jsonl"):
    if not l.startswith("{"): continue
    x = json.loads(l); c = x["config"]
    v = c.get("verified"); v = v.get("ok") if isinstance(v, dict) else (x.get("verified") if v is None else v)
    print((c.get("workload") or "")[:60], c.get("commitments_in_flight"), x["value"], x["ms_per_step"], "verified" if v else "")
PY

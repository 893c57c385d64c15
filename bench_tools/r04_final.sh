# round 4, final GPU call: the whole GPU suite, smoke(), the default line, rocprofv3 stats + PMC of the default workload, the operating points DESIGN.md section 5 quotes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_final; rm -rf $E; mkdir -p $E/profiles
( time timeout 1300 python -m pytest tests -m gpu -q ) > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $E/smoke.log 2>&1; echo "smoke rc $?" >> $E/smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $E/bench_default.json 2> $E/bench_default.err
bash bench_tools/profile.sh r04f_msm_n22 > $E/profile_n22.log 2>&1
python bench_tools/summarize_profile.py gpurun_out/prof_r04f_msm_n22 r04f_msm_n22_table 22 > $E/summarize_n22.log 2>&1
cp profiles/r04f_* $E/profiles/ 2>/dev/null; rm -rf gpurun_out/prof_r04f_msm_n22
{
  for ln in 20 22; do for pl in 1 2 3; do python bench.py --log-n $ln --pipeline $pl --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off; done; done
  python bench.py --log-n 22 --pipeline 4 --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off
  python bench.py --log-n 22 --precompute 0 --pipeline 1 --steps 10 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --verify
  python bench.py --workload fold_step --rc 900 --steps 8 --warmup 2 --verify --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --stage-ahead 1 --late-ranges 1 --verify --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --witness-ahead 2 --no-cpu-baseline --secondary 0
  python bench.py --workload compress --steps 5 --warmup 2 --verify
  python bench.py --workload store_hydrate --steps 10 --warmup 2 --verify --no-cpu-baseline
} > $E/sweep.jsonl 2> $E/sweep.err
for wl in "fold_step --rc 100 --secondary 0" "compress"; do
  name=$(echo $wl | cut -d' ' -f1)
  ( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$name -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $E/profiles/r04f_${name}_bench_under_rocprof.json 2> $E/prof_$name.err )
  cp $(ls $E/prof_$name/*/*_kernel_stats.csv | head -1) $E/profiles/r04f_${name}_kernel_stats.csv 2>/dev/null; rm -rf $E/prof_$name
done
tail -5 $E/pytest_gpu.log; cat $E/smoke.log | tail -2
python - <<PY
import json
d = json.loads([l for l in open("$E/bench_default.json") if l.startswith("{")][-1])
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for k, v in d.get("sub_records", {}).items(): print(" ", k, v.get("ms_per_step"), (v.get("config") or {}).get("verified") and "verified", v.get("wall_s"), v.get("error"))
for l in open("$E/sweep.jsonl"):
    x = json.loads(l); c = x["config"]
    print(c.get("workload", "")[:46], c.get("commitments_in_flight"), x["value"], x["ms_per_step"], c.get("verified") and "verified")
PY

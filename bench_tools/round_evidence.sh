# One GPU call that regenerates the round's measurement evidence (run through gpurun; results land in gpurun_out/evidence_<tag>/ and,
# for the rocprofv3 summaries, in gpurun_out/evidence_<tag>/profiles/ - copy those into profiles/ and commit them).
# usage: bash bench_tools/round_evidence.sh <tag>        (tag e.g. r03)
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/evidence_$TAG
rm -rf $E; mkdir -p $E/profiles
# 1. the default line exactly as the driver runs it
python bench.py --gpus 1 --steps 20 --warmup 5 > $E/bench_default.json 2> $E/bench_default.err
# 2. rocprofv3 kernel trace + stats + PMC passes of the same workload, condensed (launch classes, traffic three ways)
bash bench_tools/profile.sh ${TAG}_msm_n22 > $E/profile_n22.log 2>&1
python bench_tools/summarize_profile.py gpurun_out/prof_${TAG}_msm_n22 ${TAG}_msm_n22_table 22 > $E/summarize_n22.log 2>&1
bash bench_tools/profile.sh ${TAG}_msm_n20 --log-n 20 > $E/profile_n20.log 2>&1
python bench_tools/summarize_profile.py gpurun_out/prof_${TAG}_msm_n20 ${TAG}_msm_n20_table 20 > $E/summarize_n20.log 2>&1
cp profiles/${TAG}_* $E/profiles/ 2>/dev/null
rm -rf gpurun_out/prof_${TAG}_msm_n22 gpurun_out/prof_${TAG}_msm_n20
# 3. the sweep: every workload / operating point DESIGN.md section 5 quotes, one JSON line each
{
  for ln in 20 22; do for pl in 1 2 3; do python bench.py --log-n $ln --pipeline $pl --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off; done; done
  python bench.py --log-n 22 --dist witness --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off
  python bench.py --log-n 22 --precompute 0 --pipeline 1 --steps 10 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off
  python bench.py --gpus 2 --backend gloo --scaling strong --log-n 22 --steps 10 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off --verify   # two ranks on this box's one GPU: the strong-scaling path, functional
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --devices 0,0 --verify --no-cpu-baseline --secondary 0
  python bench.py --workload store_hydrate --steps 10 --warmup 2 --verify
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --verify
  python bench.py --workload fold_step --rc 900 --steps 10 --warmup 3 --verify --no-cpu-baseline
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --stage-ahead 1 --late-ranges 0 --verify --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --stage-ahead 1 --late-ranges 1 --verify --no-cpu-baseline --secondary 0
  python bench.py --workload compress --steps 5 --warmup 2 --verify
  python bench.py --workload poseidon_tree --steps 5 --warmup 2
  python bench.py --workload ntt --log-n 24 --steps 10 --warmup 3
  python bench.py --workload ntt --log-n 20 --steps 10 --warmup 3
} > $E/sweep.jsonl 2> $E/sweep.err
python bench_tools/small_commit_probe.py 200 > $E/small_commit_probe.jsonl 2>> $E/sweep.err
# 3b. rocprofv3 kernel stats of the other workloads (one short run each)
for wl in "fold_step --rc 100 --secondary 0" "compress" "poseidon_tree --log-n 24" "ntt --log-n 24" "store_hydrate"; do
  name=$(echo $wl | cut -d' ' -f1)
  ( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_$name -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $E/profiles/${TAG}_${name}_bench_under_rocprof.json 2> $E/prof_$name.err )
  cp $(ls $E/prof_$name/*/*_kernel_stats.csv | head -1) $E/profiles/${TAG}_${name}_kernel_stats.csv 2>/dev/null; rm -rf $E/prof_$name
done
./bench_tools/microbench > $E/profiles/${TAG}_microbench_instr_rates.txt 2>&1
# 4. store hydration: per-level kernel times (the level time is the Poseidon hash's dependency chain)
( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_hydrate -- python -m pytest tests/test_gpu_poseidon.py -q -k "hydrat" > $E/hydrate.log 2>&1 )
cp $(ls $E/prof_hydrate/*/*_kernel_stats.csv | head -1) $E/profiles/${TAG}_store_hydrate_kernel_stats.csv 2>/dev/null; rm -rf $E/prof_hydrate
# 5. the folding step's kernel timeline (hardware queue and stream per kernel; two consecutive primary-curve steps)
bash bench_tools/trace_step.sh > $E/trace_step.log 2>&1
cp gpurun_out/trace_step/timeline.txt $E/profiles/${TAG}_step_timeline_rc100.txt 2>/dev/null
tail -c 600 $E/bench_default.json; echo; wc -l $E/sweep.jsonl

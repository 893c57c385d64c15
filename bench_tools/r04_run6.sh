# round 4, GPU call 6: final suite, NTT timings with the early twist gathers, the default line of the committed code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run6; rm -rf $E; mkdir -p $E
timeout 1500 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
{
  python bench.py --workload ntt --log-n 24 --steps 20 --warmup 3 --verify
  python bench.py --workload ntt --log-n 20 --steps 20 --warmup 3 --no-cpu-baseline
  python bench.py --workload ntt --log-n 22 --steps 20 --warmup 3 --no-cpu-baseline
  python bench.py --workload ntt --log-n 17 --steps 20 --warmup 3 --no-cpu-baseline
} > $E/sweep.jsonl 2> $E/sweep.err
( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_ntt -- python bench.py --workload ntt --log-n 24 --steps 5 --warmup 2 --no-cpu-baseline > $E/ntt_bench_under_rocprof.json 2> $E/prof_ntt.err )
cp $(ls $E/prof_ntt/*/*_kernel_stats.csv | head -1) $E/r04_ntt_kernel_stats.csv 2>/dev/null; rm -rf $E/prof_ntt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $E/bench_default.json 2> $E/bench_default.err
tail -4 $E/pytest_gpu.log

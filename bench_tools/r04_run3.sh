# round 4, GPU call 3: suite after the reduce-tail fusion / XCD-aware fold kernels / batched sum-check loop, instruction-rate probes with the
# FP64-FMA multiplier, and the operating points those changes touch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run3; rm -rf $E; mkdir -p $E
timeout 1200 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
./bench_tools/microbench > $E/microbench.txt 2>&1
{
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off --sub-records off
  python bench.py --log-n 20 --pipeline 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off
  python bench.py --log-n 20 --pipeline 2 --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --verify
  python bench.py --workload fold_step --rc 900 --steps 5 --warmup 2 --no-cpu-baseline --secondary 0
  python bench.py --workload compress --steps 5 --warmup 2 --verify --no-cpu-baseline
} > $E/sweep.jsonl 2> $E/sweep.err
tail -4 $E/pytest_gpu.log; cat $E/microbench.txt | head -40

# round 4, GPU call 2: the GPU test suite (new tests included), the folding step over device lists, store hydration, the fold_vec fraction
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run2; rm -rf $E; mkdir -p $E
timeout 1200 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
{
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --verify
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --verify --devices 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --verify --devices 0,0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --verify --devices 0,0,0,0
  python bench.py --workload store_hydrate --steps 10 --warmup 2 --verify
  python bench.py --workload fold_step --rc 900 --steps 5 --warmup 2 --no-cpu-baseline --secondary 0
} > $E/sweep.jsonl 2> $E/sweep.err
tail -8 $E/pytest_gpu.log

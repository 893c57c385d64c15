# round 4, GPU call 1: the GPU test suite, the default line with its sub-records, and the window-width sweep at the step's operating point
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run1; rm -rf $E; mkdir -p $E
timeout 900 python -m pytest tests -m gpu -x -q > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $E/bench_default.json 2> $E/bench_default.err
{
  for c in 18 19 20; do
    python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --window-bits $c
    python bench.py --log-n 20 --pipeline 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off --window-bits $c
    python bench.py --log-n 20 --pipeline 2 --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off --window-bits $c
  done
} > $E/window_sweep.jsonl 2> $E/window_sweep.err
tail -5 $E/pytest_gpu.log

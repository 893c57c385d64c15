# round 4, GPU call 4: suite with the helper-key and staged-transcript tests; the step with the transcript staged inside begin; staging ahead with a helper key
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run4; rm -rf $E; mkdir -p $E
grep -m1 "model name" /proc/cpuinfo > $E/host_cpu.txt; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -i "avx512ifma\|adx\|bmi2" | tr '\n' ' ' >> $E/host_cpu.txt
timeout 1500 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $E/pytest_gpu.log
{
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --verify
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --verify --secondary 0
  python bench.py --workload fold_step --rc 900 --steps 5 --warmup 2 --no-cpu-baseline --secondary 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --verify --secondary 0 --stage-ahead 1 --late-ranges 1
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --verify --secondary 0 --stage-ahead 1 --late-ranges 1 --helper-devices 0
  python bench.py --workload fold_step --rc 100 --steps 20 --warmup 5 --no-cpu-baseline --verify --secondary 0 --devices 0,0
} > $E/sweep.jsonl 2> $E/sweep.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $E/bench_default.json 2> $E/bench_default.err
tail -4 $E/pytest_gpu.log

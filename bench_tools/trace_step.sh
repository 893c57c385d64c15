# kernel timeline of the fold-step workload (args: extra bench.py flags); output: gpurun_out/trace_step/timeline.txt (last 3 steps)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -- python bench.py --workload fold_step --steps 6 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/trace_step.log 2>&1
f=$(ls gpurun_out/trace_step/*/*_kernel_trace.csv | head -1)
python bench_tools/timeline.py $f > gpurun_out/trace_step/timeline_all.txt
python - <<PY
lines=open("gpurun_out/trace_step/timeline_all.txt").read().splitlines()
# keep the window of the last ~14 ms before the uniform-columns tail: find the last r1cs_cross_term launches
idx=[i for i,l in enumerate(lines) if "r1cs_cross_term" in l]
# the last 3 are the uniform-shape runs; the timed steps' are before them
start=idx[-6]; end=idx[-3]
open("gpurun_out/trace_step/timeline.txt","w").write("\n".join(lines[start-5:end]))
print(len(lines), start, end)
PY
rm -f $f gpurun_out/trace_step/*/*.csv
tail -2 gpurun_out/trace_step.log | cut -c1-300

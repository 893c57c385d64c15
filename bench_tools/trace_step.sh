# kernel timeline of the fold-step workload (args: extra bench.py flags).  Output: gpurun_out/trace_step/timeline.txt - two consecutive
# timed steps of the primary curve, times in microseconds relative to the first one's cross term: start, end, duration, hardware
# queue, stream, share of the duration overlapped by kernels of other queues, kernel.  NOTE: under rocprofv3 every launch call costs
# the host more (a commitment's ~45 launches block the submitting thread for > 1 ms), so the host-side gaps are wider than in a plain run;
# the device-side order and the kernel durations are what this is for.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -- python bench.py --workload fold_step --steps 6 --warmup 3 --no-cpu-baseline --secondary 0 "$@" > gpurun_out/trace_step.log 2>&1
f=$(ls gpurun_out/trace_step/*/*_kernel_trace.csv | head -1)
python bench_tools/timeline.py $f > gpurun_out/trace_step/timeline_all.txt
python - <<PY
lines = open("gpurun_out/trace_step/timeline_all.txt").read().splitlines()
# the steps' cross terms run on the folding context's own stream; the three launches at the very end (default stream) are bench.py's
# roofline leg
idx = [i for i, l in enumerate(lines) if "r1cs_cross_term_kernel<PallasFq" in l and " s  0 " not in l]
first, last = idx[-3], idx[-1]
t0 = float(lines[first].split()[0])
out = []
for l in lines[first - 12:last]:
    p = l.split()
    out.append("%9.1f %9.1f %8.1f  q%-3s s%-3s ovl %4s  %s" % (float(p[0]) - t0, float(p[1]) - t0, float(p[2]), p[5], p[7], p[9], " ".join(p[10:])))
open("gpurun_out/trace_step/timeline.txt", "w").write("\n".join(out) + "\n")
print(len(lines), len(out))
PY
rm -f $f gpurun_out/trace_step/*/*.csv
tail -1 gpurun_out/trace_step.log | cut -c1-200

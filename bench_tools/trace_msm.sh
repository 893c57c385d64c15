# kernel timeline of the pipelined MSM bench (args: extra bench.py flags) -> gpurun_out/trace_msm/timeline.txt (everything but the
# short planes/plan kernels) + per-kernel stats
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/trace_msm; mkdir -p gpurun_out/trace_msm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_msm -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --pmc off --no-plain-leg "$@" > gpurun_out/trace_msm/bench.json 2> gpurun_out/trace_msm/err.txt
f=$(ls gpurun_out/trace_msm/*/*_kernel_trace.csv | head -1)
python bench_tools/timeline.py $f | grep -v "planes_kernel\|fillBuffer\|len_\|task_base\|taskscan\|scan1\|part_start\|tasks_kernel\|copyBuffer" > gpurun_out/trace_msm/timeline.txt
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/trace_msm/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("lurk::","")[:44]
    print(f"{n:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f} tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -f gpurun_out/trace_msm/*/*_kernel_trace.csv
tail -1 gpurun_out/trace_msm/bench.json | cut -c1-200

#!/bin/bash
# quick kernel-trace only: bash bench_tools/trace.sh <tag> [bench args]
TAG=${1:-t}; shift || true
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
python - <<PY
import csv,glob
f=glob.glob("$OUT/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("lurk::","")[:44]
    print(f"{n:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY

cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_foldb -- python bench_tools/fold_bench.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/trace_foldb/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("lurk::","")[:44]
    if "r1cs" in n or "fold" in n: print(f"{n:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY

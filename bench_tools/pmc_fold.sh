#!/bin/bash
# HBM/L2 counters of the fold kernels in isolation: bash bench_tools/pmc_fold.sh
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_fold; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench_tools/fold_bench.py > $OUT/log_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench_tools/fold_bench.py > $OUT/log_write.txt 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python bench_tools/fold_bench.py > $OUT/log_tcc.txt 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch","write","tcc"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("lurk::","")[:40]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "r1cs" in k or "fold_vec" in k:
        print(k, {c: round(sum(x)/len(x)/1e3,1) for c,x in v.items()}, "(FETCH/WRITE in MB as reported KiB/1e3; TCC in K)")
PY

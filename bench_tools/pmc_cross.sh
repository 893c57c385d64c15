#!/bin/bash
# What bounds the cross term: memory-side and SQ counters of r1cs_cross_term_kernel in isolation (bench_tools/fold_bench.py), one pass per group.
# bash bench_tools/pmc_cross.sh [rc]
RC=${1:-100}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_cross; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python bench_tools/fold_bench.py $RC > $OUT/log_$i.txt 2>&1 || echo "pass $i ($grp) failed" >> $OUT/failed.txt
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("lurk::","")[:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in agg.items():
    if "r1cs_cross" in k or "fold_vec" in k:
        print(k)
        for c,x in sorted(v.items()): print("   %-32s %16.1f  (n=%d)" % (c, sum(x)/len(x), len(x)))
PY
cat $OUT/failed.txt 2>/dev/null
grep -h "ms   nnz" $OUT/log_1.txt

#!/bin/bash
# SQ counters for an arbitrary bench invocation: bash bench_tools/pmc_any.sh <tag> <kernel-substring> [bench args]
TAG=${1:-t}; KSUB=${2:-msm}; shift 2 || true
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("lurk::","")[:44]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    agg[k]["vgpr"]=[float(r["VGPR_Count"])]; agg[k]["sgpr"]=[float(r["SGPR_Count"])]; agg[k]["scratch"]=[float(r["Scratch_Size"])]
for k,v in agg.items():
    if "$KSUB" in k: print(k, {c: round(sum(x)/len(x)/ (1e6 if c not in ("dur_us","vgpr","sgpr","scratch") else 1),2) for c,x in v.items()})
PY

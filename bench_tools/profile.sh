#!/bin/bash
# Profiles `python bench.py` under rocprofv3 on the GPU box: kernel trace + stats, then the HBM
# counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage (through gpurun): bash bench_tools/profile.sh <tag> [bench.py args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --pmc off --no-plain-leg --sub-records off $*"   # --pmc off: bench.py must not start its own rocprofv3 passes under this one;
# --no-plain-leg: the plain-key sub-record launches the same accumulate kernel on 16 n entries instead of 13 n - kept out of these profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace_err.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch_err.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write_err.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq_err.txt
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -20 $f; done

#!/bin/bash
TAG=${1:-acc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 24 --warmup 3 --pmc off --no-cpu-baseline --no-plain-leg "$@" > $OUT/$name.json 2>$OUT/$name.err
  python - "$name" "$OUT/$name.json" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:30s} {r['value']:8.1f} M/s  {r['ms_per_step']:.3f} ms/step  sync {r['sync_ms_per_commit']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
run max0_r0_$rep LURK_MSM_MAX_ACC=0 LURK_MSM_ACC_R128=0 -- --pipeline 3
run max2_r0_$rep LURK_MSM_MAX_ACC=2 LURK_MSM_ACC_R128=0 -- --pipeline 3
run max2_r1_$rep LURK_MSM_MAX_ACC=2 LURK_MSM_ACC_R128=1 -- --pipeline 3
run max0_r1_$rep LURK_MSM_MAX_ACC=0 LURK_MSM_ACC_R128=1 -- --pipeline 3
run max1_w2_r0_$rep LURK_MSM_MAX_ACC=1 LURK_MSM_ACC_WAVES=2 LURK_MSM_ACC_R128=0 -- --pipeline 3
run max1_w2_r1_$rep LURK_MSM_MAX_ACC=1 LURK_MSM_ACC_WAVES=2 LURK_MSM_ACC_R128=1 -- --pipeline 3
done

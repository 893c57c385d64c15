#!/bin/bash
TAG=${1:-pr}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-plain-leg "$@" > $OUT/$name.json 2>$OUT/$name.err
  python - "$name" "$OUT/$name.json" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:34s} {r['value']:8.1f} M/s  {r['ms_per_step']:.3f} ms/step  acc {r['roofline']['avg_launch_ms']:.3f} sync {r['sync_ms_per_commit']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for R in 0 1; do for SP in 0 1 3; do for TP in 0 3; do
run r${R}_s${SP}_t${TP}_p2 LURK_MSM_ACC_R128=$R LURK_MSM_SORT_PRIO=$SP LURK_MSM_TAIL_PRIO=$TP -- --pipeline 2
done; done; done
run r1_s3_t3_p3 LURK_MSM_ACC_R128=1 -- --pipeline 3
run r1_w1_p2 LURK_MSM_ACC_R128=1 LURK_MSM_ACC_WAVES=1 -- --pipeline 2
run r1_w1_p3 LURK_MSM_ACC_R128=1 LURK_MSM_ACC_WAVES=1 -- --pipeline 3
run r0_w1_p3 LURK_MSM_ACC_R128=0 LURK_MSM_ACC_WAVES=1 -- --pipeline 3

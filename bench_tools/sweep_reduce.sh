#!/bin/bash
TAG=${1:-rd}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-plain-leg "$@" > $OUT/$name.json 2>$OUT/$name.err
  python - "$name" "$OUT/$name.json" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k=r['kernel_ms_per_commit_sync']
    print(f"{sys.argv[1]:26s} {r['value']:8.1f} M/s  {r['ms_per_step']:.3f} ms/step  sync {r['sync_ms_per_commit']:.3f}  sort {k['msm_sort']:.3f} acc {k['msm_accumulate']:.3f} fin {k['msm_finalize']:.3f} red {k['msm_reduce']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for F in 0 1; do
run f${F}_sync   LURK_MSM_REDUCE_FUSED=$F -- --pipeline 1
run f${F}_p2     LURK_MSM_REDUCE_FUSED=$F -- --pipeline 2
run f${F}_p2r128 LURK_MSM_REDUCE_FUSED=$F LURK_MSM_ACC_R128=1 -- --pipeline 2
run f${F}_p3w1   LURK_MSM_REDUCE_FUSED=$F LURK_MSM_ACC_WAVES=1 -- --pipeline 3
run f${F}_plain  LURK_MSM_REDUCE_FUSED=$F -- --pipeline 1 --precompute 0
run f${F}_n20p3  LURK_MSM_REDUCE_FUSED=$F -- --pipeline 3 --log-n 20
run f${F}_n20sync LURK_MSM_REDUCE_FUSED=$F -- --pipeline 1 --log-n 20
done

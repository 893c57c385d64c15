#!/bin/bash
# A/B of the commitments-in-flight path: accumulate launch form x stream priorities x slots (bash bench_tools/sweep_overlap.sh <tag>)
TAG=${1:-ov}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
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
for P in 2 3; do
run classic_noprio_p$P LURK_MSM_ACC_PERSISTENT=0 LURK_MSM_PRIO=0 -- --pipeline $P
run classic_prio_p$P   LURK_MSM_ACC_PERSISTENT=0 LURK_MSM_PRIO=1 -- --pipeline $P
run pers2_prio_p$P     LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=2 LURK_MSM_PRIO=1 -- --pipeline $P
run pers2_noprio_p$P   LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=2 LURK_MSM_PRIO=0 -- --pipeline $P
run pers2r128_prio_p$P LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=2 LURK_MSM_ACC_R128=1 LURK_MSM_PRIO=1 -- --pipeline $P
run pers3_prio_p$P     LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=3 LURK_MSM_PRIO=1 -- --pipeline $P
run pers3r128_prio_p$P LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=3 LURK_MSM_ACC_R128=1 LURK_MSM_PRIO=1 -- --pipeline $P
done
run n20_classic_p3 LURK_MSM_ACC_PERSISTENT=0 LURK_MSM_PRIO=0 -- --pipeline 3 --log-n 20
run n20_pers2_p3   LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=2 LURK_MSM_PRIO=1 -- --pipeline 3 --log-n 20
run n20_pers3_p3   LURK_MSM_ACC_PERSISTENT=1 LURK_MSM_ACC_WAVES=3 LURK_MSM_PRIO=1 -- --pipeline 3 --log-n 20

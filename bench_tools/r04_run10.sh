# round 4, GPU call 10: where the next witness's trace kernels are enqueued (device-side scope timelines, plain run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run10; rm -rf $E; mkdir -p $E
run() {  # label, flags...
  l=$1; shift
  LURK_PROF_TIMELINE=$E/tl_$l.txt python bench.py --workload fold_step --rc 100 --steps 8 --warmup 4 --no-cpu-baseline --secondary 0 "$@" 2>$E/$l.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'], d['config'].get('host_ms_per_step') or d.get('host_ms_per_step'))" >> $E/order.txt
}
run wa2 --witness-ahead 2
run wa1 --witness-ahead 1
run wa0 --witness-ahead 0
run stage --stage-ahead 1
run stage_nolate --stage-ahead 1 --late-ranges 0
cat $E/order.txt

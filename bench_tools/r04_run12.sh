# round 4, GPU call 12: the submit hook (next witness traced from inside begin) - tests, the step at rc = 100 / 900, both curves, timelines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run12; rm -rf $E; mkdir -p $E
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_host_cpp.py -q -x > $E/pytest_step.log 2>&1; echo "pytest rc $?" >> $E/pytest_step.log
run() {  # label, flags...
  l=$1; shift
  LURK_PROF_TIMELINE=$E/tl_$l.txt python bench.py --workload fold_step --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$E/$l.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'], d['config'].get('verified') and True, d.get('host_ms_per_step'), (d.get('secondary_curve_step') or {}).get('ms_per_step'), (d.get('both_curves') or {}))" >> $E/order.txt
}
run hook_rc100 --rc 100 --witness-ahead 3 --verify
run wa2_rc100 --rc 100 --witness-ahead 2
run hook_rc100_b --rc 100 --witness-ahead 3 --secondary 0
run wa2_rc100_b --rc 100 --witness-ahead 2 --secondary 0
run hook_rc900 --rc 900 --witness-ahead 3 --secondary 0 --steps 8 --warmup 2
run wa2_rc900 --rc 900 --witness-ahead 2 --secondary 0 --steps 8 --warmup 2
run hook_rc10 --rc 10 --witness-ahead 3 --secondary 0
run wa2_rc10 --rc 10 --witness-ahead 2 --secondary 0
cat $E/order.txt; tail -3 $E/pytest_step.log

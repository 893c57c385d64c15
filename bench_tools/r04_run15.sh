cd "${GRAFT_REPO_ROOT:-/root/repo}"
E=gpurun_out/r04_run15; rm -rf $E; mkdir -p $E
timeout 900 python -m pytest tests/test_gpu_ipa.py -q -x -k "key_fold" > $E/pytest_ipa.log 2>&1; echo "pytest rc $?" >> $E/pytest_ipa.log
tail -3 $E/pytest_ipa.log
LURK_PROF_TIMELINE=$E/tl_compress.txt python bench.py --workload compress --steps 2 --warmup 2 --no-cpu-baseline > $E/compress.json 2> $E/compress.err
python - <<PY
L=[x.split() for x in open("$E/tl_compress.txt") if not x.startswith("#")]
# last proof: from the last 'eq_evals' group start... print the final 330 scopes compactly
t_end=float(L[-1][1]); rows=[x for x in L if float(x[0])>t_end-40000]
t0=float(rows[0][0])
for x in rows: print("%9.0f %9.0f %7.0f %s" % (float(x[0])-t0, float(x[1])-t0, float(x[2]), x[4]))
PY

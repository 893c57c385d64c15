# FETCH_SIZE calibration for streaming reads and 64-byte gathers (bench_tools/fetch_calib.hip) -> gpurun_out/fetch_calib.json
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/fetch_calib; mkdir -p gpurun_out/fetch_calib
hipcc -O3 -std=c++17 --offload-arch=gfx950 bench_tools/fetch_calib.hip -o gpurun_out/fetch_calib/fetch_calib || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fetch_calib/out -- gpurun_out/fetch_calib/fetch_calib > gpurun_out/fetch_calib/known.json 2> gpurun_out/fetch_calib/err.txt
python - <<PY
import csv, glob, json
known = json.loads(open("gpurun_out/fetch_calib/known.json").read().strip().splitlines()[-1])
vals = {"calib_stream": [], "calib_gather64": []}
for f in glob.glob("gpurun_out/fetch_calib/out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in vals:
            if k in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                vals[k].append(float(r["Counter_Value"]))
out = {"known": known, "note": "FETCH_SIZE in KiB as rocprofv3 reports it; factor = known bytes / (FETCH_SIZE * 1024): what the reported figure must be multiplied by"}
for k, v in vals.items():
    if v:
        v = v[1:] if len(v) > 1 else v  # first launch touches cold pages
        mean = sum(v) / len(v)
        out[k] = {"fetch_size_kib": round(mean, 1), "launches": len(v), "factor": round(known[k + "_bytes"] / (mean * 1024.0), 4)}
json.dump(out, open("gpurun_out/fetch_calib.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/fetch_calib/out gpurun_out/fetch_calib/fetch_calib

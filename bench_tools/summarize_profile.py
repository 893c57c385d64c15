#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (bench_tools/profile.sh) into the small, tracked
artefacts under profiles/: the rocprofv3 kernel_stats table and a per-kernel PMC summary (HBM
bytes corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 128-B requests as 64 B for
coalesced reads on gfx950 -> doubled; WRITE_SIZE taken as is; both are KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("lurk::", "").strip()


def main(src, tag, log_n):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[k]["_vgpr"] = [float(r["VGPR_Count"])]
                agg[k]["_sgpr"] = [float(r["SGPR_Count"])]
                agg[k]["_lds"] = [float(r["LDS_Block_Size"])]
                agg[k]["_scratch"] = [float(r["Scratch_Size"])]
    summary = {}
    for k, v in agg.items():
        row = {c: sum(x) / len(x) for c, x in v.items()}
        row["launches"] = max(len(x) for x in v.values())
        if "FETCH_SIZE" in row or "WRITE_SIZE" in row:
            f_kib, w_kib = row.get("FETCH_SIZE", 0.0), row.get("WRITE_SIZE", 0.0)
            row["hbm_read_bytes_raw"] = f_kib * 1024
            row["hbm_read_bytes_corrected_x2"] = f_kib * 2 * 1024
            row["hbm_write_bytes"] = w_kib * 1024
            row["hbm_bytes_per_launch"] = (f_kib * 2 + w_kib) * 1024
        summary[k] = row
    with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump({"log_n": log_n, "source": src, "note": "per-launch means; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM",
                   "kernels": summary}, f, indent=1, sort_keys=True)
    acc = next((v for k, v in summary.items() if k.startswith("msm_accumulate_kernel")), None)
    if acc and "hbm_bytes_per_launch" in acc:
        with open(os.path.join(out_dir, "pmc_msm_accumulate.json"), "w") as f:
            json.dump({"log_n": log_n, "tag": tag, "hbm_bytes_per_launch": acc["hbm_bytes_per_launch"],
                       "hbm_read_bytes_raw": acc["hbm_read_bytes_raw"], "hbm_write_bytes": acc["hbm_write_bytes"]}, f)
    p = os.path.join(src, "bench_trace.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out_dir, f"{tag}_bench_under_rocprof.json"))
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))

#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (bench_tools/profile.sh) into the small, tracked artefacts under profiles/:

  <tag>_kernel_stats.csv        rocprofv3's own per-kernel table, as it is
  <tag>_launch_classes.json     the dominant kernel's launches split into the three classes a bench run mixes
                                  warm-up     empty launches of lurk_hip_msm_ctx_reserve (a few microseconds)
                                  solo        no other launch of the same kernel overlaps it in time (what `roofline.achieved` is about)
                                  overlapped  shares the device with another commitment's accumulation (the pipelined timed region)
                                with count / mean / min / max per class - rocprofv3's single "AverageNs" averages all three
  <tag>_pmc_summary.json        per-kernel PMC means per launch; HBM bytes three ways: FETCH_SIZE raw, doubled (the guide's correction for
                                wide streaming reads) and scaled by the factor bench_tools/fetch_calib.sh measured for 64-byte gathers;
                                `hbm_bytes_per_launch` takes the calibrated figure for the gather kernels and the doubled one elsewhere
usage: summarize_profile.py <prof dir> <tag> <log_n> [dominant kernel substring]"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATHER_KERNELS = ("msm_accumulate", "msm_small_kernel", "r1cs_cross_term")  # dominated by 32/64-byte gathers, not by streaming reads


def short(name):
    return name.split("(")[0].replace("void ", "").replace("lurk::", "").strip()


def gather_factor():
    """what FETCH_SIZE must be multiplied by for 64-byte gathers (bench_tools/fetch_calib.sh); None if never measured"""
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*fetch_calibration.json")), reverse=True):
        try:
            return json.load(open(p))["calib_gather64"]["factor"], os.path.basename(p)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def launch_classes(trace_csv, needle):
    """per kernel NAME containing `needle` (the plain and the persistent accumulation are different kernels): warm-up / solo / overlapped,
    where `overlapped` = another full-size launch of ANY kernel matching `needle` runs during it.  A bench run launches the same kernel on
    different work (the table key's 13 n entries, the plain key's 16 n of the `plain_sync` leg): the solo class is also listed per grid
    size, and the table-key launches are the largest group of equal grids."""
    rows = []
    with open(trace_csv) as f:
        for r in csv.DictReader(f):
            if needle in r["Kernel_Name"]:
                grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), str(grid)))
    if not rows:
        return None
    rows.sort()
    out = {"kernels": {}}
    names = sorted(set(n for _, _, n, _ in rows))
    big = {}
    for nm in names:
        dur = sorted(e - s for s, e, n, _ in rows if n == nm)
        big[nm] = statistics.median(dur[len(dur) // 2:])  # median of the upper half: a typical full launch of this kernel
    full = [(s, e, n) for s, e, n, _ in rows if e - s >= 0.1 * big[n]]

    def stat(v):
        return {"count": len(v), "mean_us": round(sum(v) / len(v) / 1e3, 2), "min_us": round(min(v) / 1e3, 2), "max_us": round(max(v) / 1e3, 2)}

    for nm in names:
        classes = collections.defaultdict(list)
        by_grid = collections.defaultdict(list)
        for s, e, n, g in rows:
            if n != nm:
                continue
            if e - s < 0.1 * big[nm]:
                classes["warm_up"].append(e - s)
                continue
            overl = any((s2, e2, n2) != (s, e, n) and s2 < e and e2 > s for s2, e2, n2 in full)
            classes["overlapped" if overl else "solo"].append(e - s)
            if not overl:
                by_grid[g].append(e - s)
        ent = {"launches": sum(len(v) for v in classes.values())}
        for k, v in classes.items():
            ent[k] = stat(v)
        if len(by_grid) > 1:
            ent["solo_by_grid_size"] = {g: stat(v) for g, v in sorted(by_grid.items(), key=lambda kv: -len(kv[1]))}
        out["kernels"][nm] = ent
    return out


def main(src, tag, log_n, needle="msm_accumulate"):
    out_dir = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if traces:
        lc = launch_classes(traces[0], needle)
        if lc:
            lc["note"] = ("rocprofv3's AverageNs for this kernel mixes the classes below; `solo` is the duration the roofline fraction is computed from "
                          "(bench.py times the same thing live with HIP events over synchronous commitments)")
            with open(os.path.join(out_dir, f"{tag}_launch_classes.json"), "w") as f:
                json.dump(lc, f, indent=1)
            print(json.dumps(lc))
    factor, factor_src = gather_factor()
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
        row = {}
        for c, x in v.items():
            full = [y for y in x if y >= 0.5 * max(x)] if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU") and max(x) > 0 else x  # drop the empty warm-up launches
            row[c] = sum(full) / len(full)
        row["launches"] = max(len(x) for x in v.values())
        if "FETCH_SIZE" in row or "WRITE_SIZE" in row:
            f_kib, w_kib = row.get("FETCH_SIZE", 0.0), row.get("WRITE_SIZE", 0.0)
            row["hbm_read_bytes_raw"] = f_kib * 1024
            row["hbm_read_bytes_x2_streaming"] = f_kib * 2 * 1024
            gather = any(g in k for g in GATHER_KERNELS)
            if gather and factor:
                row["hbm_read_bytes_calibrated_gather"] = f_kib * 1024 * factor
                row["read_correction"] = f"x{factor} (64-byte gathers, {factor_src})"
                read = f_kib * 1024 * factor
            else:
                row["read_correction"] = "x2 (wide streaming reads, MI355X_MICROARCH.md)" if not gather else "raw (gather kernel, no calibration file)"
                read = f_kib * 1024 * (1 if gather else 2)
            row["hbm_write_bytes"] = w_kib * 1024
            row["hbm_bytes_per_launch"] = read + w_kib * 1024
        summary[k] = row
    if summary:
        with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump({"log_n": log_n, "source": src, "note": "per-launch means over the full-size launches; FETCH_SIZE / WRITE_SIZE in KiB, separate passes", "kernels": summary},
                      f, indent=1, sort_keys=True)
    p = os.path.join(src, "bench_trace.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out_dir, f"{tag}_bench_under_rocprof.json"))
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), *(sys.argv[4:5]))

"""--workload msm (the default): the headline metric, Pedersen MSM Mscalar-mul/s at 2^22 on Pallas."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time

from .common import BENCH, ROOT
from .common import cpu_quota_cores, kernel_profile
from .predict import msm_predicted_value


def plain_sync_leg(args, d_bases, d_scalars, n, stream):
    """The same workload through what the literal pasta-msm drop-in does minus PCIe: a plain 64 B/point key (no
    precomputed table, nothing to amortise), one synchronous commitment at a time."""
    import torch

    import lurk_beta_amd as L

    ck = L.CommitmentKey(L.CURVE_PALLAS, d_bases, n=n, device=True, precompute=False)
    for _ in range(max(1, args.warmup)):
        ck.commit_device(d_scalars, n, is_mont=True, stream=stream)
    torch.cuda.synchronize()
    k = max(3, args.steps)
    t0 = time.perf_counter()
    for _ in range(k):
        ck.commit_device(d_scalars, n, is_mont=True, stream=stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    ck.close()
    return {"value": round(n / dt / 1e6, 3), "unit": "Mscalar-mul/s", "ms_per_commit": round(dt * 1e3, 4), "steps": k,
            "config": "plain resident key (64 B/point, 16-bit windows), synchronous: one commitment at a time, result on the host after each"}


def oneshot_leg(d_bases, d_scalars, n):
    """The literal pasta-msm drop-in, `lurk_hip_msm_pallas(out, points, npoints, scalars, is_mont)`, as an unmodified arecibo calls
    it: bases AND scalars in host memory on every call (96 B per point over PCIe), nothing resident but the library's own buffers."""
    import numpy as np

    import lurk_beta_amd as L

    B = d_bases.cpu().numpy().view(np.uint64)
    S = d_scalars.cpu().numpy().view(np.uint64)
    L.msm(L.CURVE_PALLAS, B, S, is_mont=True)  # first call allocates the cached buffers
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        L.msm(L.CURVE_PALLAS, B, S, is_mont=True)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    # the same with the opt-in key cache (the bases of the previous call stay in HBM when pointer and sampled points match)
    from lurk_beta_amd import _lib

    lib = _lib.load()
    _lib.check(lib.lurk_hip_msm_oneshot_key_cache(1))
    try:
        L.msm(L.CURVE_PALLAS, B, S, is_mont=True)
        tc = []
        for _ in range(3):
            t0 = time.perf_counter()
            L.msm(L.CURVE_PALLAS, B, S, is_mont=True)
            tc.append(time.perf_counter() - t0)
    finally:
        _lib.check(lib.lurk_hip_msm_oneshot_key_cache(0))
    dc = min(tc)
    return {"value": round(n / dt / 1e6, 3), "unit": "Mscalar-mul/s", "ms_per_call": round(dt * 1e3, 3), "pcie_bytes_per_call": 96 * n,
            "config": "host pointers in, result out, per call: H2D of scalars, sort, H2D of bases behind it, accumulate, reduce (plain 16-bit windows)",
            "with_key_cache": {"value": round(n / dc / 1e6, 3), "ms_per_call": round(dc * 1e3, 3), "pcie_bytes_per_call": 32 * n,
                               "config": "lurk_hip_msm_oneshot_key_cache(1): opt-in, the immutable key of the previous call is reused"}}


KERNEL_GROUPS = {  # kernels of one commitment, by pipeline stage (name substrings of the rocprofv3 kernel names)
    "accumulate": ("msm_accumulate", "msm_bucket_direct"),   # (the direct bucket launch: short commitments only, none at the headline size)
    "sort": ("msm_canon", "msm_hist1", "msm_scan1", "msm_part_start", "msm_scatter1", "msm_part2"),
    "plan": ("msm_taskscan", "msm_tasks_kernel", "msm_len_"),
    "finalize": ("msm_finalize", "msm_big_bucket"),
    "reduce": ("msm_planes29", "msm_reduce"),
}


def _kernel_group(name):
    for g, subs in KERNEL_GROUPS.items():
        if any(x in name for x in subs):
            return g
    return None


def collect_counters(args):
    """Three rocprofv3 --pmc passes over a short synchronous re-run of this workload (after the timed region; separate passes, as
    MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and WRITE_SIZE do not fit one): HBM bytes per launch of msm_accumulate_kernel
    (FETCH_SIZE / WRITE_SIZE in KiB) and SQ_INSTS_VALU per commitment of every pipeline stage (the input of the issue budget).
    Returns (traffic bytes per launch or None, detail or None, {stage: wave-level VALU instructions per commitment} or None)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}, None
    commits = 3  # the child runs 1 + --steps synchronous commitments
    child = [sys.executable, BENCH, "--pmc-child", "--pmc", "off", "--steps", str(commits - 1), "--no-cpu-baseline",
             "--log-n", str(args.log_n), "--dist", args.dist, "--precompute", str(args.precompute), "--window-bits", str(args.window_bits)]
    env = dict(os.environ, TMPDIR="/tmp")
    vals, valu = {}, None
    work = tempfile.mkdtemp(prefix="lurk_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            outdir = os.path.join(work, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "--"] + child
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            except Exception as e:  # noqa: BLE001
                if counter == "SQ_INSTS_VALU":
                    break
                return None, {"error": f"rocprofv3 --pmc {counter} failed: {type(e).__name__}"}, None
            per, groups = [], {}
            for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") != counter:
                            continue
                        name = r.get("Kernel_Name", "")
                        if "msm_accumulate_kernel" in name:
                            per.append(float(r["Counter_Value"]))
                        g = _kernel_group(name)
                        if g:
                            groups[g] = groups.get(g, 0.0) + float(r["Counter_Value"])
            if counter == "SQ_INSTS_VALU":
                # (the slot warm-ups of lurk_hip_msm_ctx_reserve run every kernel on 256 zero scalars: a few thousand instructions)
                valu = {g: v / commits for g, v in groups.items()} if groups else None
                continue
            if not per:
                return None, {"error": f"no {counter} rows for msm_accumulate_kernel"}, None
            # the launches of the workload itself: lurk_hip_msm_ctx_reserve warms every slot with an empty commitment, whose
            # accumulate launch moves next to nothing and would dilute a plain mean (it halved the figure once)
            full = [v for v in per if v >= 0.5 * max(per)]
            vals[counter] = (sum(full) / len(full), len(full))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fetch_kib, write_kib = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of a wide coalesced streaming read - double it - and is UNCALIBRATED for other
    # access widths: "calibrate on a known byte count in your own access pattern".  This kernel's reads are 64-byte gathers from a
    # 3.25 GiB table, so the factor comes from bench_tools/fetch_calib.sh (known bytes / reported bytes for exactly that pattern);
    # without a calibration file the raw figure is reported, and the doubled one beside it as the upper estimate.
    factor, factor_src = None, None
    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*fetch_calibration.json")), reverse=True):
        try:
            with open(pth) as fh:
                factor, factor_src = float(json.load(fh)["calib_gather64"]["factor"]), os.path.basename(pth)
            break
        except Exception:  # noqa: BLE001
            continue
    read_bytes = fetch_kib * 1024.0 * (factor if factor else 1.0)
    total = read_bytes + write_kib * 1024.0
    return total, {"source": "live: rocprofv3 --pmc over a synchronous re-run of this workload, separate passes",
                   "fetch_size_kib_raw": round(fetch_kib, 1), "write_size_kib": round(write_kib, 1), "launches": vals["FETCH_SIZE"][1],
                   "read_bytes_raw": fetch_kib * 1024.0, "read_bytes_x2_streaming_correction": 2.0 * fetch_kib * 1024.0,
                   "read_bytes_calibrated_gather": read_bytes if factor else None,
                   "correction": (f"FETCH_SIZE x {factor} - the factor {factor_src} measured for 64-byte gathers from a 3.25 GiB table (known bytes / reported bytes); "
                                  if factor else "FETCH_SIZE raw (no gather calibration file under profiles/); ") +
                                 "the guide's x2 is for wide streaming reads only and is listed beside it; WRITE_SIZE as is"}, valu


def pipeline_budget(valu_per_commit, ms_per_step):
    """The issue-cycle budget of EVERY kernel of one commitment: wave-level VALU instructions per commitment by
    stage (SQ_INSTS_VALU, this run's own PMC pass) x the stage's mean issue cost per instruction (the static mix of its ISA priced at the
    rates of profiles/r04_microbench_instr_rates.txt by bench_tools/issue_model.py mix -> profiles/rNN_issue_mix.json, the newest one is read) on 1024 SIMDs
    at 2.15 GHz.  Their sum is the floor of a commitment if everything beside the accumulation overlapped it perfectly."""
    if not valu_per_commit:
        return None
    mix, src = {}, None
    import glob

    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*issue_mix.json")), reverse=True):
        try:
            with open(pth) as fh:
                mix, src = json.load(fh), os.path.basename(pth)
            break
        except Exception:  # noqa: BLE001
            continue
    default_cpi = ACC_ISSUE_CYCLES / 2216.0
    stages, floor = {}, 0.0
    for g, insts in valu_per_commit.items():
        cpi = float(mix.get(g, {}).get("cycles_per_valu", default_cpi))
        ms = insts * cpi / (1024 * 2.15e9) * 1e3
        stages[g] = {"valu_winsts_M": round(insts / 1e6, 2), "cycles_per_valu": round(cpi, 2), "issue_ms": round(ms, 4)}
        floor += ms
    return {"stages": stages, "floor_ms_per_commit": round(floor, 4), "measured_ms_per_commit": round(ms_per_step, 4),
            "measured_over_floor": round(ms_per_step / floor, 4) if floor else None, "issue_mix": src,
            "note": "floor = sum over the stages of SQ_INSTS_VALU (live PMC pass, per commitment) x mean issue cycles per instruction of the stage's ISA / "
                    "(1024 SIMDs x 2.15 GHz): what a commitment costs if sort, plan, finalize and reduction only ever took issue slots the accumulation left"}


# Instruction-issue model of the accumulate loop (bench_tools/issue_model.py over the ISA of msm_acc.hip, profiles/r04_acc_issue_model.txt):
# per mixed addition 1226 v_mad_u64_u32 at 4.7 cycles per wave-instruction, 376 VOP3 / 64-bit / SGPR-operand instructions at 4.1
# and 614 VOP2 instructions on VGPRs, inline constants or literals at 2.3 (the rates of profiles/r04_microbench_instr_rates.txt):
# 8 716 issue cycles per wave-trip.
ACC_ISSUE_CYCLES = 1226 * 4.7 + 376 * 4.1 + 614 * 2.3


def valu_roofline(acc_ms, mixed_adds):
    # radix-2^29 XYZZ mixed addition (curve29.cuh): 8 products of 135 + 2 squarings of 99 v_mad_u64_u32, minus the
    # one reduction (54) saved by forming Y3 as a two-term lazy row
    cycles_per_wave_madd = (8 * 135 + 2 * 99 - 54) * 4.6
    peak = 1024 * 2.15e9 * 64 / cycles_per_wave_madd  # mixed additions / s if the SIMDs issued nothing but those mads
    issue_peak = 1024 * 2.15e9 * 64 / ACC_ISSUE_CYCLES  # ... if they issued the loop's whole instruction mix back to back
    ach = mixed_adds / (acc_ms * 1e-3) if acc_ms > 0 else 0.0
    return {"bound": "valu", "kernel": "msm_accumulate_kernel", "achieved": round(ach / 1e9, 3), "peak": round(peak / 1e9, 3),
            "unit": "G mixed-add/s", "frac": round(ach / peak, 4),
            "issue_model": {"peak": round(issue_peak / 1e9, 3), "frac": round(ach / issue_peak, 4), "cycles_per_wave_madd": round(ACC_ISSUE_CYCLES),
                            "note": "peak = the v_mad-only ceiling (what rounds 1-3 quoted); issue_model.peak = every instruction of the loop at its measured "
                                    "issue cost - on gfx950 only VOP2 instructions without an SGPR source issue in 2.3 cycles, every VOP3 / 64-bit / SGPR-operand "
                                    "form takes 4.1: profiles/r04_acc_issue_model.txt, profiles/r04_microbench_instr_rates.txt"}}


def msm_window_bits(args, n):
    """the library's own choice for a key of n points (msm.hip: set_bases_device)"""
    if args.window_bits:
        return args.window_bits
    if not args.precompute:
        return 16
    if n <= 1 << 16:
        return 8 if n <= 1 << 14 else 6  # the small-commitment form
    return 16 if n <= 1 << 18 else 20


def msm_windows(args, n):
    return -(-256 // msm_window_bits(args, n))


def cpu_baseline(args, gpu_result):
    """The CPU leg: oracle/msm_fast.c - a pasta-msm-shaped Pippenger (4 x 64 Montgomery on mulx/adcx, Booth windows, XYZZ
    buckets, (window, chunk) tiles over all cores) - on the SAME workload at the same size when it fits the time bound
    (2^22 takes well under 10 s on the GPU box's host), timed at the OpenMP default and at every logical CPU, best kept.
    A port (the reference's pasta-msm cannot be built here: no Rust), so "kind": "port"."""
    import numpy as np

    from oracle import coracle as C

    log_m = min(args.cpu_sample_log_n, args.log_n)
    m = 1 << log_m
    dist_id = 0 if args.dist == "uniform" else 1
    B = C.synth_bases(0, m)
    S = C.synth_scalars(1, 1, dist_id, m)
    C.msm_fast(0, B[:4096], S[:4096])  # warm up the thread pool
    quota = cpu_quota_cores()
    logical = os.cpu_count() or 1
    if quota:  # a cgroup CPU quota caps the useful thread count whatever the host has: sweep around it
        candidates = sorted({max(1, int(quota)), max(1, int(quota * 1.5)), max(1, int(quota * 2))})
    else:
        candidates = sorted({C.lib().orc_num_threads(), logical})
    best = None
    for threads in candidates:
        info = {}
        t0 = time.perf_counter()
        r = C.msm_fast(0, B, S, nthreads=threads, info=info)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, threads, info, r)
    dt, threads, info, r = best
    few = max(1, min(8, int(quota) if quota else 8))  # per-core rate from a run no quota throttles
    t0 = time.perf_counter()
    C.msm_fast(0, B[: m // 4], S[: m // 4], nthreads=few)
    per_core = (m // 4) / (time.perf_counter() - t0) / 1e6 / few
    out = {
        "value": round(m / dt / 1e6, 4),
        "unit": "Mscalar-mul/s",
        "cores": threads,
        "threads_used": threads,
        "host_cores": os.cpu_count(),
        "cpu_quota_cores": quota,
        "per_core_value": round(per_core, 4),
        "per_core_note": f"Mscalar-mul/s per thread from a {few}-thread run on a quarter of the points (no throttling); a full unthrottled host scales this by its "
                         "physical core count at best",
        "kind": "port",
        "sample": f"{'the same' if log_m == args.log_n else 'the first'} 2^{log_m} points of the workload, one MSM, {dt:.2f} s; oracle/msm_fast.c "
                  f"(pasta-msm-shaped Pippenger: mulx Montgomery, Booth {info.get('window_bits')}-bit windows, XYZZ buckets, {info.get('tiles')} tiles); "
                  "NOT the reference's pasta-msm binary",
        "field_mul_ns_single_core": round(C.fast_mul_ns(0, 1_000_000), 1),
    }
    if gpu_result is not None:
        import lurk_beta_amd as L

        out["matches_gpu_result"] = bool(C.jac_to_affine(0, r) == L.point_to_affine(L.CURVE_PALLAS, gpu_result))
    return out


def msm_workload(args, lib, world, rank):
    """The headline: K complete commitments C = sum_i s_i ck_i over Pallas, scalars and key resident in HBM, `--pipeline` of them in
    flight; the K-step region is repeated `--reps` times (each bracketed by a barrier + device synchronisation on both sides, starting
    and ending with an empty pipeline) and the MEDIAN region is reported, as criterion's flat sampling does for
    /root/reference/benches/fibonacci.rs:144-166 (one 88 ms region moves by +-2-4 % from box to box and run to run)."""
    import statistics

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    import lurk_beta_amd as L
    from lurk_beta_amd import synth
    from lurk_beta_amd.distributed import allreduce_commitment

    from .sub_records import any_failed, sub_records, summarize

    n = 1 << args.log_n
    if args.scaling == "strong":  # ONE commitment of 2^log_n points over all ranks
        assert n % world == 0, "strong scaling: 2^log_n must divide by the number of ranks"
        n //= world
    dist_id = 0 if args.dist == "uniform" else 1
    first = rank * n  # rank r owns points [r*n, (r+1)*n) of the global commitment
    d_bases = synth.bases(L.CURVE_PALLAS, n, first=first)
    d_scalars = synth.scalars(L.FIELD_PALLAS_FQ, 1, dist_id, n, first=first, mont=True)
    torch.cuda.synchronize()
    t_setup = time.perf_counter()
    depth = max(1, min(6, args.pipeline))  # LURK_MSM_SLOTS
    ck = L.CommitmentKey(L.CURVE_PALLAS, d_bases, n=n, device=True, precompute=bool(args.precompute), window_bits=args.window_bits)
    ck.reserve(n, depth)  # every slot's workspace is part of the once-per-key setup, not of whichever step touches the slot first
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t_setup) * 1e3
    stream = torch.cuda.current_stream().cuda_stream

    def finish(part):
        if world == 1:
            return part
        # the path's one exchange: all_gather of the 96-byte partial commitments (RCCL over xGMI), then
        # the group sum on every rank
        return allreduce_commitment(L.CURVE_PALLAS, part)

    def run_steps(k):
        """k complete commitments; with depth > 1 up to `depth` of them are in flight at once."""
        res = None
        if depth == 1:
            for _ in range(k):
                res = finish(ck.commit_device(d_scalars, n, is_mont=True, stream=stream))  # 96-byte Jacobian, host
            return res
        for i in range(k):
            slot = i % depth
            if i >= depth:
                res = finish(ck.wait(slot))
            ck.submit_device(slot, d_scalars, n, is_mont=True, stream=stream)
        for i in range(max(0, k - depth), k):
            res = finish(ck.wait(i % depth))
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:  # under rocprofv3 --pmc: a few synchronous commitments, no timing, no JSON line
        for _ in range(1 + args.steps):
            ck.commit_device(d_scalars, n, is_mont=True, stream=stream)
        torch.cuda.synchronize()
        ck.close()
        return 0

    # part of the once-per-process setup, like the key and its slots: the device is brought to its steady clocks with ~0.4 s of the same
    # commitments before the W warm-up steps (the first process on a fresh box measured up to 5 % low without it: 20 steps are 90 ms)
    # (a FIXED number of commitments: with N > 1 every step is a collective, so all ranks must run the same count)
    t_dev = time.perf_counter()
    device_warmup_commitments = max(8, min(512, (96 << 22) >> args.log_n))
    run_steps(device_warmup_commitments)
    device_warmup_ms = (time.perf_counter() - t_dev) * 1e3
    result = run_steps(args.warmup)
    lib.lurk_hip_profile_enable(1)
    lib.lurk_hip_profile_reset()
    reps = max(1, args.reps)
    regions = []
    for _ in range(reps):  # every repetition is the contract's region: barrier + synchronize, EXACTLY K steps, barrier + synchronize
        sync()
        t0 = time.perf_counter()
        result = run_steps(args.steps)
        sync()
        regions.append(time.perf_counter() - t0)
    # what ran inside the regions: with commitments in flight the accumulation is the PERSISTENT kernel on the slots' low-priority streams
    timed_acc = {k: kernel_profile(lib, k) for k in ("msm_accumulate_persistent", "msm_accumulate")}
    # per-kernel durations (HIP events on the launch stream) come from synchronous commitments so that
    # overlapping launches of the other slots do not stretch them
    for _ in range(2):  # the synchronous path's own warm-up (first use of slot 0 on the caller's stream)
        ck.commit_device(d_scalars, n, is_mont=True, stream=stream)
    lib.lurk_hip_profile_reset()
    t1 = time.perf_counter()
    nsync = 5
    for _ in range(nsync):
        ck.commit_device(d_scalars, n, is_mont=True, stream=stream)
    sync_ms = (time.perf_counter() - t1) / nsync * 1e3
    lib.lurk_hip_profile_enable(0)
    if world > 1:  # the job is as slow as its slowest rank, repetition by repetition
        tmax = torch.tensor(regions, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        regions = [float(x) for x in tmax.tolist()]
    elapsed = statistics.median(regions)

    kernels = {k: kernel_profile(lib, k) for k in ("msm_sort", "msm_tasks", "msm_accumulate", "msm_finalize", "msm_reduce")}

    rc = 0
    if rank == 0:
        total_points = n * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_points / (elapsed / args.steps) / 1e6
        acc_ms, acc_cnt = kernels["msm_accumulate"]
        acc_avg_ms = acc_ms / max(acc_cnt, 1)
        alg_bytes = 96.0 * n  # per launch: one rank's shard
        achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
        # the kernel the timed regions launched: its mean launch under overlap, how many were resident on average
        # (sum of launch durations / wall time of the regions) and the share of one commitment (overlapped mean / residency)
        p_ms, p_cnt = timed_acc["msm_accumulate_persistent"]
        a_ms, a_cnt = timed_acc["msm_accumulate"]  # prefix match: persistent + plain launches
        wall_ms = sum(regions) * 1e3
        if p_cnt:
            resident = p_ms / wall_ms
            timed_kernel = {"kernel": "msm_accumulate_persistent_kernel", "launches": int(p_cnt), "mean_ms_overlapped": round(p_ms / p_cnt, 4),
                            "mean_resident": round(resident, 3), "ms_per_commitment_share": round(p_ms / p_cnt / max(resident, 1e-9), 4),
                            "achieved_GBps_share": round(alg_bytes / (p_ms / p_cnt / max(resident, 1e-9) * 1e-3) / 1e9, 3),
                            "source": "library profiler (HIP events on the slot's accumulate stream), all repetitions of the timed region"}
        else:
            timed_kernel = {"kernel": "msm_accumulate_kernel", "launches": int(a_cnt), "mean_ms_overlapped": round(a_ms / max(a_cnt, 1), 4),
                            "source": "library profiler (HIP events on the launch stream), all repetitions of the timed region"}
        # HBM traffic of the dominant kernel and the VALU instruction counts of every stage, measured by THIS run: three rocprofv3 --pmc
        # passes over a short synchronous re-run of the same workload (after the timed region).  When that is not possible (no
        # rocprofv3, N > 1, --pmc off) traffic stays null.
        traffic, traffic_detail, valu_insts = None, None, None
        if args.pmc == "auto" and world == 1:
            traffic, traffic_detail, valu_insts = collect_counters(args)
        rv = valu_roofline(acc_avg_ms, msm_windows(args, n) * n)
        rv["pipeline"] = pipeline_budget(valu_insts, ms_per_step)
        out = {
            "metric": "MSM Mscalar-mul/s (Pallas Pedersen commitment, bases+scalars resident in HBM)",
            "value": round(value, 3),
            "unit": "Mscalar-mul/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            # written down before any N-GPU run exists (bench_workloads/predict.py): single-GPU pieces only
            "predicted_value": round(msm_predicted_value(args.log_n, world, args.scaling, depth), 1) if args.precompute and args.dist == "uniform" else None,
            "dtype": "u32x8 (255-bit Montgomery, integer VALU)",
            "data": "synthetic",
            "config": {
                "workload": (f"2^{args.log_n}-point Pallas Pedersen MSM per GPU ({args.dist} scalars), " if args.scaling == "weak" else
                             f"ONE 2^{args.log_n}-point Pallas Pedersen MSM cut across {world} GPU(s), {n} points per GPU ({args.dist} scalars), ") +
                            f"{'precomputed-table' if args.precompute else 'plain'} resident commitment key",
                "points_per_gpu": n,
                "total_points": total_points,
                "window_bits": msm_window_bits(args, n),
                "parallelism": f"shard{world}+all_gather(96B)" if world > 1 else "single",
                "commitments_in_flight": depth,
                # disclosed beside --warmup (verdict, round 5): these run BEFORE the W declared warm-up steps, untimed, once per process
                "device_warmup_commitments": device_warmup_commitments,
                "device_warmup_ms_once": round(device_warmup_ms, 1),
                "timed_region": f"median of {reps} repetitions of the {args.steps}-step region (each: barrier + synchronize, {args.steps} commitments, barrier + synchronize)",
                "ms_per_step_by_repetition": [round(r / args.steps * 1e3, 4) for r in regions],
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "msm_accumulate_kernel",
                "achieved": round(achieved, 3),
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6),
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "avg_launch_ms": round(acc_avg_ms, 4),
                "avg_launch_source": f"{nsync} synchronous commitments after the timed region (plain launch, nothing else on the device)",
                "timed_region_kernel": timed_kernel,
                "algorithmic_bytes_per_launch": alg_bytes,
                "mixed_additions_per_launch": msm_windows(args, n) * n,
                "note": "integer-VALU bound (v_mad_u64_u32 issue), not HBM bound: see roofline_valu and DESIGN.md",
            },
            # the honest ceiling for this kernel is VALU issue, not HBM: a mixed addition needs 1224 v_mad_u64_u32
            # (4.6 cycles per wave-instruction per SIMD, measured: profiles/r01_microbench_instr_rates.txt) on
            # 1024 SIMDs at the ~2.15 GHz the chip sustains here; shifts/masks/lazy adds come on top
            "roofline_valu": rv,
            "kernel_ms_per_commit_sync": {k: round(v[0] / max(v[1], 1) * (v[1] / nsync), 4) for k, v in kernels.items()},
            "sync_ms_per_commit": round(sync_ms, 4),
            "setup_ms_once": round(setup_ms, 1),
            "device_warmup_ms_once": round(device_warmup_ms, 1),
        }
        if not args.no_plain_leg and world == 1 and args.precompute:
            out["plain_sync"] = plain_sync_leg(args, d_bases, d_scalars, n, stream)
        if not args.no_plain_leg and world == 1:
            out["oneshot_host_pointers"] = oneshot_leg(d_bases, d_scalars, n)
        if args.verify:
            # sum_i s_i [k_i]G == [sum_i s_i k_i] G over ALL ranks' points (bases have known discrete logs)
            from oracle import coracle as C

            k = C.synth_base_scalars(0, total_points)
            sc = C.synth_scalars(1, 1, dist_id, total_points)
            want = C.jac_to_affine(0, C.gen_mul(0, C.dot(1, k, sc)))
            out["verified"] = bool(L.point_to_affine(L.CURVE_PALLAS, result) == want)
            if not out["verified"]:
                rc = 1
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, result if args.cpu_sample_log_n >= args.log_n else None)
            if out["cpu_baseline"].get("matches_gpu_result") is False:
                rc = 1
        if args.sub_records == "auto" and world == 1:
            # the metric's own workload (one folding step at rc = 100, both curve halves) and the two other named kernels at their
            # BASELINE sizes, each verified against the oracle, on the same clock as this line (the key's 3.5 GiB go back first)
            ck.close()
            del d_bases, d_scalars
            torch.cuda.empty_cache()
            subs = sub_records(args)
            out["sub_records"] = subs
            # the compact view twice: inside `config` (a record that keeps the parsed config keeps it) and as the LAST key of the line
            summary = summarize(subs)
            out["config"]["sub_summary"] = summary
            out["sub_summary"] = summary
            if any_failed(summary):
                rc = 1
        print(json.dumps(out), flush=True)
        if rc:
            print("bench.py: a parity check or a sub-record failed (see `verified`, `cpu_baseline.matches_gpu_result`, `sub_summary`)", file=sys.stderr, flush=True)
    ck.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc

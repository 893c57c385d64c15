"""--workload poseidon_tree / ntt: the other two named kernels of the path at their BASELINE sizes."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time

from .common import BENCH, ROOT, cpu_leg_omp, cpu_leg_threads


def poseidon_mads_per_hash(arity):
    """v_mad_u64_u32 per hash of the kernels' schedule (poseidon29.cuh), radix-2^29 layer: product 135, squaring 99, one lazy row
    of k terms 81 k + 54.  Full round: t S-boxes (2 squarings + 1 product) + t rows of t terms; partial round: 1 S-box + one
    row of t terms + t - 1 products; canonical in (arity products) and out (1)."""
    import lurk_beta_amd as L

    t = arity + 1
    rf, rp = L.poseidon_constants(L.FIELD_PALLAS_FQ, arity)[:2]  # the product's own round numbers (lurk_hip_poseidon_constants: host code)
    sbox, row = 2 * 99 + 135, 81 * t + 54
    return rf * (t * sbox + t * row) + rp * (sbox + row + (t - 1) * 135) + (arity + 1) * 135


def poseidon_valu_roofline(arity, hashes, kernel_ms):
    mads = poseidon_mads_per_hash(arity)
    peak = 1024 * 2.15e9 * 64 / (mads * 4.6)  # hashes / s if the SIMDs issued nothing but those mads (4.6 cycles per wave-instruction, measured)
    ach = hashes / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    return {"bound": "valu", "kernel": "poseidon_batch_kernel", "achieved": round(ach / 1e6, 2), "peak": round(peak / 1e6, 2), "unit": f"M hash{arity}/s",
            "frac": round(ach / peak, 4), "mads_per_hash": mads}


def other_workloads(args, lib, world, rank):
    """Poseidon arity-8 tree (BASELINE configs[2]: 2^24 Pallas-Fq leaves) and the radix-2 NTT, same timing
    contract: inputs resident in HBM, K timed steps.  N > 1: the tree is ONE tree sharded by subtrees with a
    single 8 x 32-byte all-gather (SURVEY.md section 8e, strong scaling); the NTT runs replicas."""
    import numpy as np
    import torch
    import torch.distributed as dist

    cpu_threads, cpu_info = cpu_leg_threads()
    if rank == 0 and (args.verify or not args.no_cpu_baseline):
        cpu_leg_omp(cpu_threads)  # every CPU leg below (the --verify run doubles as the baseline) with the thread count it reports

    import lurk_beta_amd as L
    from lurk_beta_amd import _lib, synth

    stream = torch.cuda.current_stream().cuda_stream
    F = L.FIELD_PALLAS_FQ
    if args.workload == "poseidon_tree":
        log_n = args.log_n if args.log_n % 3 == 0 else 24
        n = 1 << log_n
        scaling, parallelism = "weak", "single"
        if world == 1:
            d_leaves = synth.scalars(F, 2, 0, n)
            d_levels = torch.empty(((n - 1) // 7, 4), dtype=torch.int64, device="cuda")

            def step():
                _lib.check(lib.lurk_hip_poseidon_tree8_dev(F, _lib.ptr(d_leaves), n, _lib.ptr(d_levels), _lib.ptr(stream)))

            per_step_units = n
        else:
            # ONE tree of n leaves; the 8 subtrees below the root are dealt to the ranks, each rank
            # reduces its subtrees, the 8 x 32-byte roots are all-gathered (RCCL) and hashed once more everywhere
            assert world in (2, 4, 8) and n >= 64, "an arity-8 tree shards over 2, 4 or 8 ranks"
            scaling, parallelism = "strong", f"subtrees{world}"
            per_rank, sub = 8 // world, n // 8
            d_leaves = synth.scalars(F, 2, 0, per_rank * sub, first=rank * per_rank * sub)
            d_levels = [torch.empty(((sub - 1) // 7, 4), dtype=torch.int64, device="cuda") for _ in range(per_rank)]
            d_roots = torch.empty((per_rank, 4), dtype=torch.int64, device="cuda")
            d_all = torch.empty((8, 4), dtype=torch.int64, device="cuda")
            d_root = torch.empty((1, 4), dtype=torch.int64, device="cuda")

            def step():
                for j in range(per_rank):
                    _lib.check(lib.lurk_hip_poseidon_tree8_dev(F, _lib.ptr(d_leaves[j * sub:]), sub, _lib.ptr(d_levels[j]), _lib.ptr(stream)))
                    d_roots[j].copy_(d_levels[j][-1])
                if args.backend == "nccl":
                    dist.all_gather_into_tensor(d_all, d_roots)
                else:
                    parts = [torch.empty((per_rank, 4), dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(parts, d_roots.cpu())
                    d_all.copy_(torch.cat(parts))
                _lib.check(lib.lurk_hip_poseidon_batch_dev(F, 8, _lib.ptr(d_all), 1, _lib.ptr(d_root), _lib.ptr(stream)))

            per_step_units = n / world  # the value line multiplies by world: n leaves per step in total

        unit, kname = "Mleaves/s", "poseidon_batch"
        alg_bytes = (32.0 * n + 64.0 * ((n - 1) // 7)) / world  # leaves read once; every internal node written once and read once
        workload = f"Poseidon arity-8 tree over 2^{log_n} Pallas-Fq leaves ({(n - 1) // 7} hash8)"
    else:
        log_n = args.log_n
        n = 1 << log_n
        d_data = synth.scalars(F, 3, 0, n)

        def step():
            _lib.check(lib.lurk_hip_ntt_dev(F, _lib.ptr(d_data), log_n, 0, _lib.ptr(stream)))

        scaling, parallelism = "weak", "single" if world == 1 else f"replicas{world}"
        unit, per_step_units, kname = "Melements/s", n, "ntt"
        if log_n >= 12:  # wave-resident passes of <= 8 stages, bit reversal folded into the first (ntt.hip)
            passes = (log_n + 7) // 8
        else:            # small sizes: bit-reversal pass + one LDS pass
            passes = 2
        alg_bytes = 64.0 * passes * n
        workload = f"radix-2 NTT, 2^{log_n} Pallas-Fq elements, {passes} passes over memory (parity unpinned: no reference counterpart)"
        ntt_mults = n * (log_n / 2.0 + (passes - 1) + 2)  # butterflies + twists between passes + conversion in and out
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    lib.lurk_hip_profile_enable(1)
    lib.lurk_hip_profile_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    lib.lurk_hip_profile_enable(0)
    if world > 1:  # the job is as slow as its slowest rank
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    verified, cpu_full = None, None
    if args.verify and args.workload == "poseidon_tree" and rank == 0:
        # the whole tree again on the CPU (oracle/oracle.c, every core): the root must be the same 32 bytes.  2^24 leaves are
        # 2 396 745 hash8 - tens of seconds of host time, outside the timed region; the same run is the CPU baseline below
        from oracle import coracle as C

        leaves = C.synth_scalars(1, 2, 0, n)
        t1 = time.perf_counter()
        want = [int(x) for x in np.asarray(C.poseidon_tree8(1, leaves)).reshape(-1)[:4]]
        cpu_full = time.perf_counter() - t1
        got_t = d_levels[-1] if world == 1 else d_root[0]
        got = [int(x) for x in got_t.cpu().numpy().view(np.uint64).reshape(-1)[:4]]
        assert got == want, "tree root differs from the oracle"
        verified = {"ok": True, "against": f"oracle/oracle.c: the whole 2^{log_n}-leaf tree recomputed on the CPU, root compared", "oracle_s": round(cpu_full, 1)}
    if args.verify and args.workload == "ntt" and rank == 0:
        # one forward transform of the workload's input, every element against the oracle's textbook NTT (parity unpinned upstream)
        from oracle import coracle as C

        d_chk = synth.scalars(F, 3, 0, n)
        _lib.check(lib.lurk_hip_ntt_dev(F, _lib.ptr(d_chk), log_n, 0, _lib.ptr(stream)))
        torch.cuda.synchronize()
        host = C.synth_scalars(1, 3, 0, n)
        t1 = time.perf_counter()
        want = C.ntt(1, host)
        cpu_full = time.perf_counter() - t1
        assert np.array_equal(d_chk.cpu().numpy().view(np.uint64).reshape(-1, 4), want), "NTT output differs from the oracle"
        verified = {"ok": True, "against": f"oracle/oracle.c: forward NTT of the same 2^{log_n} elements, all outputs compared", "oracle_s": round(cpu_full, 1)}
        del d_chk
    tot, cnt = ctypes.c_double(), ctypes.c_uint64()
    _lib.check(lib.lurk_hip_profile_get(kname.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
    if rank == 0:
        kernel_ms_per_step = tot.value / args.steps
        achieved = alg_bytes / (kernel_ms_per_step * 1e-3) / 1e9 if kernel_ms_per_step > 0 else 0.0
        out = {
            "metric": f"{args.workload} throughput", "value": round(per_step_units * world / (elapsed / args.steps) / 1e6, 3), "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u32x8 (255-bit Montgomery, integer VALU)",
            "data": "synthetic", "config": {"workload": workload, "parallelism": parallelism, "verified": verified},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": None, "kernel_ms_per_step": round(kernel_ms_per_step, 4),
                         "algorithmic_bytes_per_step": alg_bytes},
        }
        if args.workload == "poseidon_tree":
            out["roofline_valu"] = poseidon_valu_roofline(8, ((n - 1) // 7) / world, kernel_ms_per_step)
        if args.workload == "ntt":
            # HIP-event time of every pass (library profiler, un-profiled run): the figure to hold against a rocprofv3 kernel summary of
            # the same command - rocprofv3's kernel trace stretched this kernel by ~9 % in round 4 (0.75 + 2 x 0.86 = 2.48 ms per
            # transform in profiles/r04_ntt_kernel_stats.csv against 2.27-2.29 un-profiled)
            per_pass = []
            for k in range(4):
                t_ms, t_n = ctypes.c_double(), ctypes.c_uint64()
                _lib.check(lib.lurk_hip_profile_get(f"pass_ntt_{k}".encode(), ctypes.byref(t_ms), ctypes.byref(t_n)))
                if t_n.value:
                    per_pass.append(round(t_ms.value / t_n.value, 4))
            out["roofline"]["per_pass_ms_hip_events"] = per_pass
            out["roofline"]["per_pass_frac_of_hbm"] = [round(64.0 * n / (x * 1e-3) / 8e12, 4) for x in per_pass if x > 0]
        if args.workload == "ntt" and kernel_ms_per_step > 0:
            # the honest ceiling: field products on the radix-2^29 layer (135 v_mad_u64_u32 each at 4.6 cycles per wave-instruction,
            # 1024 SIMDs, ~2.15 GHz: profiles/r01_microbench_instr_rates.txt), not HBM
            peak = 1024 * 2.15e9 * 64 / (135 * 4.6)
            ach = ntt_mults / (kernel_ms_per_step * 1e-3)
            out["roofline_valu"] = {"bound": "valu", "kernel": "ntt_wave_pass_kernel", "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2),
                                    "unit": "G field-mul/s", "frac": round(ach / peak, 4), "field_muls_per_step": ntt_mults}
        if not args.no_cpu_baseline:
            from oracle import coracle as C

            what = "leaves" if args.workload == "poseidon_tree" else "elements"
            if cpu_full is not None:  # --verify has just run the whole workload on the CPU: that run is the baseline
                m, dt, sample_desc = n, cpu_full, f"the whole workload (2^{log_n} {what}), the --verify run"
            else:
                m = min(n, 1 << 18)
                sample = C.synth_scalars(1, 2 if args.workload == "poseidon_tree" else 3, 0, m)
                t1 = time.perf_counter()
                if args.workload == "poseidon_tree":
                    C.poseidon_tree8(1, sample)
                else:
                    C.ntt(1, sample)
                dt = time.perf_counter() - t1
                sample_desc = f"first 2^18 {what} of the same workload"
            out["cpu_baseline"] = {"value": round(m / dt / 1e6, 4), "unit": unit, "cores": cpu_threads, **cpu_info, "kind": "port",
                                   "sample": f"{sample_desc}, {dt:.2f} s (oracle/oracle.c, OpenMP, {cpu_threads} threads)"}
        print(json.dumps(out), flush=True)

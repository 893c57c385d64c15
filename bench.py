#!/usr/bin/env python3
"""bench.py - Pedersen MSM throughput of the HIP hot path (BASELINE.json metric: "MSM Mscalar-mul/s").

A "step" is one commitment  C = sum_i s_i * ck_i  over Pallas: scalars and the commitment key are
already resident in HBM when the timed region starts (the PCIe-inclusive rate is noted in
DESIGN.md, never here).  By default three commitments are in flight (`--pipeline 3`, a context's three async slots:
the sort of the third runs under the two resident accumulations; 8 of 8 paired runs at K = 20 were faster and steadier
than two in flight, 958-971 against 900-959 Mscalar-mul/s): every step is still one complete MSM and all K results
are produced inside the timed region, which starts and ends with an empty pipeline (so a small K pays the fill and
drain: K = 20 / 40 measure about 4.35 / 4.25 ms per step); `--pipeline 1` is the fully synchronous form, and
the default line also carries `sync_ms_per_commit`, the plain-key synchronous sub-record and the one-shot
host-pointer sub-record.  The resident key
carries the precomputed per-window table by default (`--precompute 0` = plain 64 B/point key).  N = 1: n = 2^log_n points on one GPU (default 2^22, the size the metric
is quoted at; `--log-n 20` is BASELINE.json configs[1]).  N > 1: one process per GPU, every rank
owns its own 2^log_n-point shard of an (N * 2^log_n)-point commitment (weak scaling); each step
ends with the path's real exchange: an RCCL all_gather of the 96-byte partial commitments and the
group sum on every rank.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (msm_accumulate): algorithmic
bytes (96 B/point: 32 B scalar + 64 B base) over its mean launch duration measured with HIP events on
the launch stream inside the timed region.  `cpu_baseline` times the CPU oracle (a port, not the
reference: the reference cannot be built here) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--dist", choices=["uniform", "witness"], default="uniform")
    ap.add_argument("--precompute", type=int, default=1,
                    help="1 = resident key with the per-window precomputed table (13 x 64 B per point, 20-bit windows); 0 = plain key")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="commitments in flight (1 = synchronous; 2..3 = async slots: the tail of one overlaps the accumulation of the next)")
    ap.add_argument("--stage-ahead", type=int, default=0,
                    help="fold_step: 1 = the next step's witness is traced and its commitment started one step ahead (lurk_hip_fold_step_prefetch); "
                         "2 = the same from inside begin (the submit hook calls prefetch: the staged commitment runs in the background class beside commit(T)); "
                         "0 = plain begin (default: measured 4.4 ms against 4.05 ms staged whole / 5.1 ms staged with late ranges at rc = 100, DESIGN.md)")
    ap.add_argument("--witness-ahead", type=int, default=3,
                    help="fold_step: the witness of step k+1 is traced while step k is in progress (the reference's producer thread): 3 = enqueued from the step's submit hook "
                         "(lurk_hip_fold_ctx_set_submit_hook: behind the step's opening kernels, beside its commitments; default: 3.5 ms at rc = 100); 2 = enqueued after begin "
                         "has returned, so that it runs while the host derives r (3.85); 1 = enqueued ahead of this step's commitments (4.2); 0 = traced at the start of its own step")
    ap.add_argument("--secondary", type=int, default=1, help="fold_step: 1 = also time the secondary-curve (Vesta, ~10^4 constraints) half of a step")
    ap.add_argument("--late-ranges", type=int, default=1, help="fold_step with --stage-ahead: 1 = 12 000 positions of W2 arrive with begin (the augmented circuit's), 0 = none")
    ap.add_argument("--ipa-resident-key", type=int, default=1, help="compress: 1 = inner-product rounds under the resident key (composed scalars), 0 = fold the key")
    ap.add_argument("--window-bits", type=int, default=0, help="window-bit override for the precomputed-table mode (16..20)")
    ap.add_argument("--workload", choices=["msm", "poseidon_tree", "ntt", "fold_step", "compress", "store_hydrate"], default="msm",
                    help="msm = the headline metric; poseidon_tree / ntt = the other hot-path kernels (BASELINE configs[2], N1); "
                         "fold_step = synthetic stand-in for one Nova folding step of benches/fibonacci.rs (configs[0]/[3])")
    ap.add_argument("--rc", type=int, default=100, help="fold_step: reduction count (frames per step), 100 or 900")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the "
                                                      "N > 1 path on a single-GPU box: every rank then shares GPU 0)")
    ap.add_argument("--verify", action="store_true",
                    help="rank 0 checks the workload's result against the oracle after the timed loop: msm - the discrete-log checksum; fold_step - one more step, "
                         "every output vs oracle.c / msm_fast.c / the oracle transcript; compress - the oracle's verifier on the last proof; poseidon_tree - the CPU tree")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-n", type=int, default=22)
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto",
                    help="auto = after the timed region, re-run two synchronous commitments under `rocprofv3 --pmc` (FETCH_SIZE and "
                         "WRITE_SIZE in separate passes) and report the dominant kernel's HBM traffic per launch in roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-plain-leg", action="store_true", help="skip the plain-key synchronous sub-record (plain_sync)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="msm with --gpus N: weak = every rank its own 2^log_n points (the default the driver's --gpus sweep runs); strong = ONE commitment of "
                         "2^log_n points cut across the ranks (2^log_n / N points per rank) - what a folding step's commitment does on N GPUs")
    ap.add_argument("--devices", default="",
                    help="fold_step: comma-separated device list, e.g. 0,0 or 0,1,2,3: the commitment key is cut across these devices inside ONE process "
                         "(lurk_hip_msm_multi_* + lurk_hip_fold_ctx_create_multi); a repeated id puts several slices on one GPU (functional and overhead check)")
    ap.add_argument("--helper-devices", default="",
                    help="fold_step with --stage-ahead 1: comma-separated devices that each hold a copy of the commitment key and commit the instances staged "
                         "ahead in turn (lurk_hip_fold_ctx_add_helper: staging ahead across GPUs); e.g. 1,2,3 on a node, 0 on a one-GPU box (functional)")
    ap.add_argument("--sub-records", choices=["auto", "off"], default="auto",
                    help="auto = the default msm line at N = 1 also carries the other workloads of the path as verified sub-records "
                         "(fold_step_rc100, poseidon_tree_2_24, ntt_2_24, compress_2_20: each a child run of this file with --verify, same --steps / --warmup)")
    args = ap.parse_args()

    # N > 1 without a launcher: become the launcher (one rank per GPU, the same command line the driver uses)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or drop WORLD_SIZE")

    import numpy as np
    import torch
    import torch.distributed as dist

    import lurk_beta_amd as L
    from lurk_beta_amd import _lib, synth
    from lurk_beta_amd.distributed import allreduce_commitment

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > torch.cuda.device_count() and args.backend == "nccl":
        sys.exit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s): RCCL needs one GPU per rank "
                 "(--backend gloo shares GPU 0 between the ranks: functional check only)")
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    lib = _lib.load()
    _lib.check(lib.lurk_hip_set_device(dev))
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.workload == "fold_step":
        return fold_step_workload(args, lib, world, rank)
    if args.workload == "compress":
        return compress_workload(args, lib, world, rank)
    if args.workload == "store_hydrate":
        return store_hydrate_workload(args, lib, world, rank)
    if args.workload != "msm":
        return other_workloads(args, lib, world, rank)

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
    depth = max(1, min(4, args.pipeline))
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
        return

    # part of the once-per-process setup, like the key and its slots: the device is brought to its steady clocks with ~0.4 s of the same
    # commitments before the W warm-up steps (the first process on a fresh box measured up to 5 % low without it: 20 steps are 90 ms)
    # (a FIXED number of commitments: with N > 1 every step is a collective, so all ranks must run the same count)
    t_dev = time.perf_counter()
    run_steps(max(8, min(512, (96 << 22) >> args.log_n)))
    device_warmup_ms = (time.perf_counter() - t_dev) * 1e3
    result = run_steps(args.warmup)
    lib.lurk_hip_profile_enable(1)
    lib.lurk_hip_profile_reset()
    sync()
    t0 = time.perf_counter()
    result = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
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
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    def prof(prefix):
        tot, cnt = ctypes.c_double(), ctypes.c_uint64()
        _lib.check(lib.lurk_hip_profile_get(prefix.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
        return tot.value, cnt.value

    kernels = {k: prof(k) for k in ("msm_sort", "msm_tasks", "msm_accumulate", "msm_finalize", "msm_reduce")}

    if rank == 0:
        total_points = n * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_points / (elapsed / args.steps) / 1e6
        acc_ms, acc_cnt = kernels["msm_accumulate"]
        acc_avg_ms = acc_ms / max(acc_cnt, 1)
        alg_bytes = 96.0 * n  # per launch: one rank's shard
        achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
        # HBM traffic of the dominant kernel, measured by THIS run: two rocprofv3 --pmc passes over a short synchronous
        # re-run of the same workload (after the timed region).  When that is not possible (no rocprofv3, N > 1, --pmc off)
        # traffic stays null and the last committed profile is quoted under its own name and tag.
        traffic, traffic_detail = None, None
        if args.pmc == "auto" and world == 1:
            traffic, traffic_detail = collect_traffic(args)
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
                "algorithmic_bytes_per_launch": alg_bytes,
                "mixed_additions_per_launch": msm_windows(args, n) * n,
                "note": "integer-VALU bound (v_mad_u64_u32 issue), not HBM bound: see roofline_valu and DESIGN.md",
            },
            # the honest ceiling for this kernel is VALU issue, not HBM: a mixed addition needs 1224 v_mad_u64_u32
            # (4.6 cycles per wave-instruction per SIMD, measured: profiles/r01_microbench_instr_rates.txt) on
            # 1024 SIMDs at the ~2.15 GHz the chip sustains here; shifts/masks/lazy adds come on top
            "roofline_valu": valu_roofline(acc_avg_ms, msm_windows(args, n) * n),
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
            assert out["verified"], "commitment does not match the discrete-log checksum"
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, result if args.cpu_sample_log_n >= args.log_n else None)
        if args.sub_records == "auto" and world == 1:
            # the metric's own workload (one folding step at rc = 100, both curve halves) and the two other named kernels at their
            # BASELINE sizes, each verified against the oracle, on the same clock as this line (the key's 3.5 GiB go back first)
            ck.close()
            del d_bases, d_scalars
            torch.cuda.empty_cache()
            out["sub_records"] = sub_records(args)
        print(json.dumps(out), flush=True)
    ck.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _frame_structured_columns(rng, row_of_entry, num_cons, num_vars, num_io):
    """Column pattern of the Lurk step circuit: W = [globals | frame 0 aux | frame 1 aux | ...] (frames are
    synthesized independently and their aux concatenated, /root/reference/src/lem/multiframe.rs:699-702, 11 141
    constraints and 9 119 aux per frame, src/lem/eval.rs:1966-1967), so frame f's rows touch frame f's block (88 %),
    the globals at the front (6 %), the previous frame's block (4 %: its outputs) and the constant-one column u (2 %)."""
    import numpy as np

    nf = max(1, num_cons // 11141)
    cons_pf, vars_pf = -(-num_cons // nf), max(1, num_vars // nf)
    frame = np.minimum(row_of_entry // cons_pf, nf - 1)
    kind = rng.random(row_of_entry.size)
    local = frame * vars_pf + rng.integers(0, vars_pf, row_of_entry.size)
    prev = np.maximum(frame - 1, 0) * vars_pf + rng.integers(0, vars_pf, row_of_entry.size)
    glob = rng.integers(0, min(256, num_vars), row_of_entry.size)
    cols = np.where(kind < 0.88, local, np.where(kind < 0.94, glob, np.where(kind < 0.98, prev, num_vars)))
    return np.minimum(cols, num_vars + num_io).astype(np.uint64)


def synth_r1cs_shape(field_id, p, num_cons, num_vars, num_io, seed=7, uniform_columns=False):
    """Synthetic CSR triple shaped like the Lurk step circuit (3-4 entries per row, one row in 300 a 255-entry
    bit decomposition, coefficients mostly +-1 / small): the bench's own generator (numpy), values in Montgomery form."""
    import numpy as np

    rng = np.random.default_rng(seed)
    ncols = num_vars + 1 + num_io
    table_ints = [1, p - 1, 2, p - 2, 3, 4, 8, 16, 256, 1 << 32, p - (1 << 16)] + [int(rng.integers(1, 1 << 62)) ** 4 % p for _ in range(21)]
    table = np.array([[(v << 256) % p >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)] for v in table_ints], dtype=np.uint64)
    weights = np.array([40, 25, 5, 2, 2, 1, 1, 1, 1, 1, 1] + [1] * 21, dtype=np.float64)
    weights /= weights.sum()

    def sparse(one_per_row=False):
        cnt = np.ones(num_cons, dtype=np.uint64) if one_per_row else rng.integers(3, 5, num_cons).astype(np.uint64)
        if not one_per_row:
            cnt[rng.integers(0, num_cons, max(1, num_cons // 300))] = min(ncols, 255)
        indptr = np.zeros(num_cons + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        rows = np.repeat(np.arange(num_cons, dtype=np.int64), cnt.astype(np.int64))
        if one_per_row:
            indices = np.full(nnz, num_vars, dtype=np.uint64)
        elif uniform_columns:
            indices = rng.integers(0, ncols, nnz).astype(np.uint64)
        else:
            indices = _frame_structured_columns(rng, rows, num_cons, num_vars, num_io)
        data = np.ascontiguousarray(table[rng.choice(len(table_ints), size=nnz, p=weights)])
        return indptr, indices, data

    return sparse(), sparse(), sparse(one_per_row=True)


def fold_step_workload(args, lib, world, rank):
    """Synthetic stand-in for the device work of ONE Nova folding step of benches/fibonacci.rs on the primary (Pallas) curve
    (the bench itself needs cargo + arecibo and cannot run here), through the step entry points:
      W2 assembled in HBM: 14 hash4 + 6 hash8 + 1 commitment + 3 bit-decomposition slot blocks per frame written by the trace
        kernels (lurk_hip_slot_witness_dev), the non-slot remainder of every frame (1 311 aux, what the CPU synthesis produces)
        copied in over PCIe                                                      (src/lem/multiframe.rs:520-592, 699-702)
      lurk_hip_fold_step_begin: commit(W2) || cross term T over the step circuit's rows || commit(T)   (nova.rs:287-293)
      lurk_hip_fold_step_finish(r): [W | u | X] <- z1 + r z2, E <- E1 + r T
    Sizes on Pallas: 8 951 aux per frame (7 640 slot aux: bit decompositions are 298 instead of BN254's 354; + 1 311) and
    10 973 constraints per frame (11 141 - 3 x 56), from src/lem/eval.rs:1960-1967 and multiframe.rs:495-497.
    Reported as "equivalent Lurk iterations/s" = rc / t(step).  Left out: what stays on the CPU in the reference (the transcript,
    circuit synthesis of the frame bodies, the small secondary-curve fold): an upper bound on the end-to-end rate, flagged synthetic."""
    import numpy as np
    import torch

    import lurk_beta_amd as L
    from lurk_beta_amd import _lib, synth

    rc = args.rc
    F = L.FIELD_PALLAS_FQ
    mf = L.MultiFrameWitness(F, rc, globals_len=64, body_len=1311)
    n_w, n_t, n_io = mf.w_len, 10973 * rc, 6
    n_key = max(n_w, n_t)
    q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
    stream = torch.cuda.current_stream().cuda_stream
    d_bases = synth.bases(L.CURVE_PALLAS, n_key)
    pre = {"hash4": synth.scalars(F, 3, 1, 14 * rc * 4, mont=True), "hash8": synth.scalars(F, 4, 1, 6 * rc * 8, mont=True),
           "commitment": synth.scalars(F, 5, 1, rc * 3, mont=True), "bit_decomp": synth.scalars(F, 6, 1, 3 * rc, mont=True)}
    globals_pinned = torch.empty((mf.globals_len, 4), dtype=torch.int64).pin_memory()   # (a pageable source would make the async copy wait for the stream)
    globals_pinned.copy_(synth.scalars(F, 7, 1, mf.globals_len, mont=True).cpu())
    globals_host = globals_pinned.numpy().view(np.uint64)
    bodies_host = torch.empty((rc, mf.body_len, 4), dtype=torch.int64).pin_memory()
    bodies_host.copy_(synth.scalars(F, 8, 1, rc * mf.body_len, mont=True).reshape(rc, mf.body_len, 4).cpu())
    bodies_np = bodies_host.numpy().view(np.uint64)
    d_w2s = [torch.zeros((n_w, 4), dtype=torch.int64, device="cuda") for _ in range(2)]
    d_w2 = d_w2s[0]
    x2 = synth.scalars(F, 9, 0, n_io, mont=True).cpu().numpy().view(np.uint64)
    t_setup = time.perf_counter()
    host_mats = synth_r1cs_shape(F, q, n_t, n_w, n_io)
    shape = L.R1CSShape(F, n_t, n_w, n_io, *host_mats)
    shape_setup_s = time.perf_counter() - t_setup
    if not args.verify:
        host_mats = None
    info = shape.info()
    # the challenge of every step is derived by the library's transcript (arecibo's PoseidonRO over pp_digest, U1, U2, comm_T: 128 bits),
    # on the host between begin and finish, as NIFS::prove does
    pp_digest = 0x1F3C5A7990B2D4E6F8123456789ABCDEF0FEDCBA9876543210AA55AA55AA55
    r_chal = 0x0FEDCBA0987654321234567890ABCDEF  # the secondary-curve leg below still feeds a constant of that size
    r_mont = np.array([((r_chal << 256) % q) >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)
    last_r = [r_mont]
    torch.cuda.synchronize()
    devices = [int(x) for x in args.devices.split(",")] if args.devices else None
    if devices:  # the key cut across a device list inside this process: slices commit concurrently, 96-byte partials summed on the host
        assert not args.stage_ahead, "--devices: staging ahead is not available with a multi-device key"
        ck = L.MultiCommitmentKey(L.CURVE_PALLAS, d_bases.cpu().numpy().view(np.uint64), devices, precompute=bool(args.precompute), window_bits=args.window_bits)
    else:
        ck = L.CommitmentKey(L.CURVE_PALLAS, d_bases, n=n_key, device=True, precompute=bool(args.precompute), window_bits=args.window_bits)
    ctx = L.FoldingContext(L.CURVE_PALLAS, shape, ck)
    ctx.set_pp_digest(pp_digest)
    helper_keys = []
    if args.helper_devices:
        assert args.stage_ahead and not devices, "--helper-devices goes with --stage-ahead 1 and a single-device key"
        for hd in [int(x) for x in args.helper_devices.split(",")]:
            _lib.check(lib.lurk_hip_set_device(hd))
            with torch.cuda.device(hd):
                hb = d_bases if d_bases.device.index == hd else d_bases.to(f"cuda:{hd}")
                hk = L.CommitmentKey(L.CURVE_PALLAS, hb, n=n_key, device=True, precompute=bool(args.precompute), window_bits=args.window_bits)
                hk.reserve(n_key, 3)
            helper_keys.append(hk)
            ctx.add_helper(hk)
        _lib.check(lib.lurk_hip_set_device(torch.cuda.current_device()))
    z1 = synth.scalars(F, 1, 1, n_w + 1 + n_io, mont=True).cpu().numpy().view(np.uint64)   # a running instance with witness-like values
    e1 = synth.scalars(F, 2, 0, n_t, mont=True).cpu().numpy().view(np.uint64)              # a running error vector (uniform, like any folded T)
    ident = np.zeros(12, dtype=np.uint64)
    # the running instance's commitments are the commitments of the running vectors (what RecursiveSNARK::verify re-computes)
    if devices:
        ctx.set_running(z1, e1, ck.commit(z1[:n_w], is_mont=True), ck.commit(e1, is_mont=True))
    else:
        ctx.set_running(z1, e1, ck.commit_device(torch.from_numpy(z1[:n_w].view(np.int64)).cuda(), n_w, is_mont=True),
                        ck.commit_device(torch.from_numpy(e1.view(np.int64)).cuda(), n_t, is_mont=True))

    # --stage-ahead 1: the step circuit's range of the NEXT witness is traced and its commitment started before this
    # step opens (lurk-beta synthesizes witnesses ahead of the folding loop, nova.rs:304-326); the augmented circuit's own
    # variables depend on the previous fold and arrive with begin: modelled as the first 9 000 and the last 3 000 positions
    if not devices:
        ck.reserve(n_key, 4)
    lo, hi = (9000, n_w - 3000) if args.stage_ahead and args.late_ranges else (0, n_w)
    late_host = synth.scalars(F, 10, 1, lo + n_w - hi, mont=True).cpu().numpy().view(np.uint64)
    patches = [(0, late_host[:lo]), (hi, late_host[lo:])] if lo else []
    staged_k = [0]

    def stage():
        buf = d_w2s[staged_k[0] & 1]
        staged_k[0] += 1
        mf.assemble(buf, pre, globals_host, bodies_np, mont=True, stream=stream)     # W2 in HBM (slot traces on the device)
        ctx.prefetch(buf[lo:hi], lo, stream=stream)                                  # commit(step circuit's range) starts now

    phase = {"assemble_and_stage": 0.0, "begin": 0.0, "transcript": 0.0, "finish": 0.0}  # host wall time per call site (begin blocks on the commitments)

    def step():
        t_a = time.perf_counter()
        if args.stage_ahead:
            if args.stage_ahead == 1:
                stage()                                                               # the next step's, under this step's work
            t_b = time.perf_counter()
            cw, ct = ctx.begin_prefetched(x2, patches)                               # late ranges + cross term + commit(T) (2: + stage() from the submit hook)
        elif args.witness_ahead:
            # the witness of step k+1 is produced while step k folds (lurk-beta's producer thread, nova.rs:304-326); this step's W2 was
            # produced a step ago.  --witness-ahead 1: its trace kernels are enqueued BEFORE this step's commitments and run beside
            # them; 2: AFTER begin has returned, so that they run while the host derives r (the device is idle there)
            k = staged_k[0]
            staged_k[0] += 1
            if args.witness_ahead == 1:
                mf.assemble(d_w2s[(k + 1) & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[(k + 1) & 1].cuda_stream)
            t_b = time.perf_counter()
            if args.witness_ahead == 3:  # traced from the step's submit hook: behind the step's opening kernels, beside its commitments
                hook_k[0] = k + 1
            cw, ct = ctx.begin(d_w2s[k & 1], x2, stream=wstreams[k & 1].cuda_stream)  # both commitments + the cross term
            if args.witness_ahead == 2:
                mf.assemble(d_w2s[(k + 1) & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[(k + 1) & 1].cuda_stream)
        else:
            mf.assemble(d_w2, pre, globals_host, bodies_np, mont=True, stream=stream)
            t_b = time.perf_counter()
            cw, ct = ctx.begin(d_w2, x2, stream=stream)                              # both commitments + the cross term
        t_c = time.perf_counter()
        r = ctx.challenge()  # r = RO(pp_digest, U1, U2, comm_T): U1 and U2 were absorbed inside begin, one permutation is left (lurk_hip_fold_step_challenge)
        t_r = time.perf_counter()
        ctx.finish(r)
        last_r[0] = r
        t_d = time.perf_counter()
        phase["assemble_and_stage"] += t_b - t_a
        phase["begin"] += t_c - t_b
        phase["transcript"] += t_r - t_c
        phase["finish"] += t_d - t_r
        return cw, ct

    if args.stage_ahead:
        stage()
        if args.stage_ahead == 2:  # the next instance is traced, staged and its commitment started from inside begin (the submit hook)
            ctx.set_submit_hook(stage)
    elif args.witness_ahead:
        wstreams = [torch.cuda.Stream(), torch.cuda.Stream()]  # witness k is produced on stream k & 1, into buffer k & 1
        hook_k = [0]
        if args.witness_ahead == 3:
            ctx.set_submit_hook(lambda: mf.assemble(d_w2s[hook_k[0] & 1], pre, globals_host, bodies_np, mont=True, stream=wstreams[hook_k[0] & 1].cuda_stream))
        mf.assemble(d_w2s[0], pre, globals_host, bodies_np, mont=True, stream=wstreams[0].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    lib.lurk_hip_profile_enable(1)
    lib.lurk_hip_profile_reset()
    torch.cuda.synchronize()
    for k in phase:
        phase[k] = 0.0
    import gc
    gc.collect()  # (as timeit does: no cyclic-garbage collection inside the timed region)
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    # (the folds are ordered on the context's own stream; torch.cuda.synchronize() is device-wide)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    lib.lurk_hip_profile_enable(0)
    if args.stage_ahead == 2 or (not args.stage_ahead and args.witness_ahead == 3):
        ctx.set_submit_hook(None)
    if args.stage_ahead:  # drain the instance staged by the last timed step (one was staged before the region: K stagings inside it)
        ctx.begin_prefetched(x2, patches)
        ctx.finish(r_mont)
    verified = None
    if args.verify and rank == 0:
        verified = verify_fold_step(L, ctx, host_mats, F, q, n_w, n_t, n_io, d_bases, pp_digest, x2,
                                    lambda buf: mf.assemble(buf, pre, globals_host, bodies_np, mont=True, stream=stream), d_w2s[0])

    def kernel_ms(name):
        tot, cnt = ctypes.c_double(), ctypes.c_uint64()
        _lib.check(lib.lurk_hip_profile_get(name.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
        return tot.value / max(cnt.value, 1), cnt.value

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        nnz = sum(info["nnz"])
        ct_ms, _ = kernel_ms("r1cs_cross_term")
        fv_ms, _ = kernel_ms("fold_vec")
        tr_ms, tr_n = kernel_ms("poseidon_trace")
        bd_ms, _ = kernel_ms("bit_decomp_trace")
        acc_ms, acc_n = kernel_ms("msm_accumulate")       # mean launch over the timed region (HIP events on its launch stream): 2 per step
        acc_bytes = 96.0 * (n_w + n_t) / 2.0              # algorithmic bytes of the mean launch: 32 B scalar + 64 B base per point
        # algorithmic HBM bytes of the cross-term kernel: 8 B per CSR record + 4 B per row pointer, two 32-byte gathers
        # per record (z1, z2), 32 B of T per row; fold_vec: two reads + one write of 32 B per element
        ct_bytes = nnz * 8.0 + 3 * 4.0 * n_t + 2 * 32.0 * nnz + 32.0 * n_t
        fv_bytes = 96.0 * ((n_w + 1 + n_io) + n_t) / 2
        res = {
            "metric": "equivalent Lurk iterations/s (synthetic stand-in for one Nova folding step, Pallas)",
            "value": round(rc / (ms * 1e-3), 1), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if devices else "weak", "vs_baseline": None,
            "dtype": "u32x8 (255-bit Montgomery, integer VALU)", "data": "synthetic",
            "config": {"staged_ahead": args.stage_ahead, "witness_ahead": 0 if args.stage_ahead else args.witness_ahead,
                       "devices": devices, "distinct_devices": len(set(devices)) if devices else 1,
                       "helper_devices": args.helper_devices or None,
                       "workload": f"fold-step stand-in rc={rc} through lurk_hip_fold_step_{'prefetch/begin_prefetched' if args.stage_ahead else 'begin'}/finish: W2 ({n_w} aux: {21 * rc} Poseidon + {3 * rc} bit-decomposition "
                                   f"slot blocks traced on the device + {rc} x 1311 body aux over PCIe) -> MSM(W2) + cross term over {n_t} rows ({nnz} non-zeros, "
                                   f"{info['distinct_coefficients']} distinct coefficients) + MSM(T) -> fold of [W|u|X] and E",
                       "note": "device work + the transcript (r derived per step by the library's PoseidonRO on the host); body synthesis not modelled; "
                               "the secondary-curve half is the separate secondary_curve_step record",
                       "verified": verified,
                       "r1cs_columns": "frame-structured (88 % frame-local, 6 % globals, 4 % previous frame, 2 % u): a builder-chosen model of the step circuit's sparsity, "
                                       "see fold_kernels.r1cs_cross_term_uniform_columns for the structure-free case",
                       "shape_setup_s_once": round(shape_setup_s, 2)},
            "host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in phase.items()},
            # the step's dominant kernel is the bucket accumulation of its two commitments; the cross term (fold_kernels below) is the HBM-side one
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": round(acc_bytes / (acc_ms * 1e-3) / 1e9, 3) if acc_ms else None,
                         "peak": 8000.0, "unit": "GB/s", "frac": round(acc_bytes / (acc_ms * 1e-3) / 8e12, 6) if acc_ms else None, "traffic": None,
                         "avg_launch_ms": round(acc_ms, 4), "launches_per_step": acc_n // max(args.steps, 1), "algorithmic_bytes_per_launch": acc_bytes,
                         "note": "96 B per point over the mean of the step's two commitments (W2 and T), launches timed inside the step (they share the device with "
                                 "the cross term and each other's sort); integer-VALU bound as in the msm workload: see its roofline_valu"},
            "fold_kernels": {
                "r1cs_cross_term": {"ms": round(ct_ms, 4), "algorithmic_bytes": ct_bytes, "achieved_GBps": round(ct_bytes / (ct_ms * 1e-3) / 1e9, 1) if ct_ms else None,
                                    "hbm_frac": round(ct_bytes / (ct_ms * 1e-3) / 8e12, 4) if ct_ms else None},
                "fold_vec": {"ms_per_launch": round(fv_ms, 4), "algorithmic_bytes_per_launch": fv_bytes,
                             "achieved_GBps": round(fv_bytes / (fv_ms * 1e-3) / 1e9, 1) if fv_ms else None,
                             "hbm_frac": round(fv_bytes / (fv_ms * 1e-3) / 8e12, 4) if fv_ms else None},
                "slot_witness_trace": {"poseidon_ms_per_launch": round(tr_ms, 4), "launches_per_step": tr_n // max(args.steps, 1),
                                       "bit_decomp_ms_per_launch": round(bd_ms, 4), "bytes_written_per_step": mf.slots_len * rc * 32.0},
            },
        }
        # the secondary-curve half of the step (Vesta, scalars in Fp): arecibo's augmented circuit on the other curve of the cycle is
        # ~10^4 constraints; its NIFS::prove runs BEFORE the primary's in prove_step and the two depend on each other through the
        # circuits, so a whole step is the sum.  W2 comes from host memory here (that circuit is synthesized on the CPU).
        if args.secondary:
            P_MOD = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
            nc2, nv2, nio2 = 10_000, 10_000, 2
            shape2 = L.R1CSShape(L.FIELD_PALLAS_FP, nc2, nv2, nio2, *synth_r1cs_shape(L.FIELD_PALLAS_FP, P_MOD, nc2, nv2, nio2, seed=11, uniform_columns=True))
            ck2 = L.CommitmentKey(L.CURVE_VESTA, synth.bases(L.CURVE_VESTA, max(nc2, nv2)), n=max(nc2, nv2), device=True, precompute=bool(args.precompute))
            ck2.reserve(max(nc2, nv2), 3)
            ctx2 = L.FoldingContext(L.CURVE_VESTA, shape2, ck2)
            ctx2.set_running(synth.scalars(L.FIELD_PALLAS_FP, 31, 1, nv2 + 1 + nio2, mont=True).cpu().numpy().view(np.uint64),
                             synth.scalars(L.FIELD_PALLAS_FP, 32, 0, nc2, mont=True).cpu().numpy().view(np.uint64), ident, ident)
            w2_sec = torch.empty((nv2, 4), dtype=torch.int64).pin_memory()
            w2_sec.copy_(synth.scalars(L.FIELD_PALLAS_FP, 33, 1, nv2, mont=True).cpu())
            w2_sec_np = w2_sec.numpy().view(np.uint64)
            x2_sec = synth.scalars(L.FIELD_PALLAS_FP, 34, 0, nio2, mont=True).cpu().numpy().view(np.uint64)
            r_mont2 = np.array([((r_chal << 256) % P_MOD) >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)
            for _ in range(3):
                ctx2.begin(w2_sec_np, x2_sec)
                ctx2.finish(r_mont2)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            reps2 = max(args.steps, 10)
            for _ in range(reps2):
                ctx2.begin(w2_sec_np, x2_sec)
                ctx2.finish(r_mont2)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t2) / reps2 * 1e3
            res["secondary_curve_step"] = {"ms_per_step": round(ms2, 4), "curve": "vesta", "constraints": nc2, "variables": nv2,
                                           "note": "arecibo's augmented circuit on the secondary curve is ~10^4 constraints [SURVEY 8: MEM]; two latency-bound "
                                                   "commitments of 10^4 points + cross term + folds, W2 from host memory"}
            res["both_curves_ms_per_step"] = round(ms + ms2, 4)
            res["both_curves_iterations_per_s"] = round(rc / ((ms + ms2) * 1e-3), 1)
            ctx2.close()
            ck2.close()
            shape2.close()
        # the same cross term over a structure-free shape (uniformly random columns): the other end of the sparsity range
        lib.lurk_hip_profile_enable(1)
        lib.lurk_hip_profile_reset()
        shape_u = L.R1CSShape(F, n_t, n_w, n_io, *synth_r1cs_shape(F, q, n_t, n_w, n_io, uniform_columns=True))
        d_z1 = torch.from_numpy(z1.view(np.int64)).cuda()
        d_t = torch.empty((n_t, 4), dtype=torch.int64, device="cuda")
        for _ in range(3):
            shape_u.cross_term(d_z1, torch.cat([d_w2, d_z1[n_w:]]), out=d_t, stream=stream)
        torch.cuda.synchronize()
        cu_ms, _ = kernel_ms("r1cs_cross_term")
        lib.lurk_hip_profile_enable(0)
        shape_u.close()
        res["fold_kernels"]["r1cs_cross_term_uniform_columns"] = {"ms": round(cu_ms, 4), "hbm_frac": round(ct_bytes / (cu_ms * 1e-3) / 8e12, 4) if cu_ms else None}
        if not args.no_cpu_baseline:
            from oracle import coracle as C

            m = min(n_t, 1 << 22)
            B = C.synth_bases(0, m)
            s_w, s_t = C.synth_scalars(1, 1, 1, min(n_w, m)), C.synth_scalars(1, 2, 0, m)
            C.msm_fast(0, B[:4096], s_t[:4096])
            t1 = time.perf_counter()
            C.msm_fast(0, B[: min(n_w, m)], s_w)
            C.msm_fast(0, B, s_t)
            dt = time.perf_counter() - t1
            scale = (n_w + n_t) / (min(n_w, m) + m)
            res["cpu_baseline"] = {"value": round(rc / (dt * scale), 2), "unit": "iterations/s", "cores": C.lib().orc_num_threads(), "kind": "port",
                                   "sample": f"the step's two MSMs ({min(n_w, m)} and {m} points{'' if scale == 1 else ', scaled linearly to the full sizes'}) in {dt:.2f} s with oracle/msm_fast.c "
                                             "(pasta-msm-shaped Pippenger, all cores); fold arithmetic, witness generation and transcript not included"}
        print(json.dumps(res), flush=True)
    ctx.close()
    for hk in helper_keys:
        hk.close()
    ck.close()
    shape.close()


def verify_fold_step(L, ctx, host_mats, F, q, n_w, n_t, n_io, d_bases, pp_digest, x2, assemble, d_w2):
    """--verify: ONE more step after the timed loop through lurk_hip_fold_step, every output against the oracle at the bench's own size:
    comm_W2 and comm_T (oracle/msm_fast.c), r (the oracle's transcript over the oracle's instance), T and the folded (z, E) element by
    element (oracle/oracle.c), the folded instance's commitments.  The checker only: nothing here is timed."""
    import numpy as np
    import torch

    from oracle import coracle as C
    from oracle import pyref as R

    f, curve = 1, 0
    t0 = time.perf_counter()
    mats = [(ip, ix, C.from_mont(f, d)) for ip, ix, d in host_mats]
    bases = d_bases.cpu().numpy().view(np.uint64).reshape(-1, 8)
    z1m, e1m = ctx.read()
    z1, e1 = C.from_mont(f, z1m), C.from_mont(f, e1m)
    cw1, ce1, _, _ = ctx.instance()
    pt = lambda a: None if a == (0, 0) else a
    cw1_o = pt(C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_w], z1[:n_w])))   # the running instance's commitments, recomputed
    ce1_o = pt(C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_t], e1)))
    ok = {"running_comm_W": pt(L.point_to_affine(curve, cw1)) == cw1_o, "running_comm_E": pt(L.point_to_affine(curve, ce1)) == ce1_o}
    assemble(d_w2)
    torch.cuda.synchronize()
    w2 = C.from_mont(f, d_w2.cpu().numpy().view(np.uint64).reshape(-1, 4))
    x2c = C.from_mont(f, x2.reshape(-1, 4))
    cw, ct, r_mont = ctx.step(d_w2, x2, pp_digest, stream=torch.cuda.current_stream().cuda_stream)
    z2 = np.concatenate([w2, C.ints_to_limbs([1]), x2c])
    u1 = C.limbs_to_ints(z1[n_w:n_w + 1])[0]
    t = C.cross_term(f, *[C.spmv(f, *M, z1) for M in mats], *[C.spmv(f, *M, z2) for M in mats], u1, 1)
    cw2_o, ct_o = C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_w], w2)), C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_t], t))
    ok["comm_W2"] = L.point_to_affine(curve, cw) == cw2_o
    ok["comm_T"] = L.point_to_affine(curve, ct) == ct_o
    r = R.nifs_challenge("pallas", pp_digest, cw1_o, ce1_o, u1, C.limbs_to_ints(z1[n_w + 1:]), pt(cw2_o), C.limbs_to_ints(x2c), pt(ct_o))
    ok["challenge"] = C.limbs_to_ints(C.from_mont(f, r_mont.reshape(1, 4)))[0] == r
    zf, ef = C.axpy(f, z1, z2, r), C.axpy(f, e1, t, r)
    gz, ge = ctx.read()
    ok["folded_z"] = bool(np.array_equal(C.from_mont(f, gz), zf))
    ok["folded_E"] = bool(np.array_equal(C.from_mont(f, ge), ef))
    gcw, gce, _, _ = ctx.instance()
    ok["folded_comm_W"] = L.point_to_affine(curve, gcw) == C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_w], zf[:n_w]))
    ok["folded_comm_E"] = L.point_to_affine(curve, gce) == C.jac_to_affine(curve, C.msm_fast(curve, bases[:n_t], ef))
    if not all(ok.values()):
        raise SystemExit(f"bench.py --verify: the fold step does not match the oracle: {ok}")
    return {"ok": True, "checks": sorted(ok), "oracle_s": round(time.perf_counter() - t0, 1),
            "against": "oracle/oracle.c (spmv, cross term, axpy), oracle/msm_fast.c (6 commitments), oracle/pyref.py (transcript), one extra step after the timed loop"}


def compress_workload(args, lib, world, rank):
    """Stand-in for the primary-curve half of CompressedSNARK::prove (/root/reference/src/proof/nova.rs:341-356) at the rc = 100 step
    circuit's padded size (2^20 constraints, 2^20 variables): lurk_beta_amd/spartan.py - outer + inner + batching sum-checks, the
    transposed sparse mat-vec, one inner-product-argument opening over a 2^20-point key (20 rounds of key folding + 2 MSMs each) -
    every vector resident in HBM, the SHA3 transcript and a few field elements per round on the host.  Functional, not
    byte-compatible with arecibo (oracle/spartan_ref.py explains); a "step" is one whole proof."""
    import numpy as np
    import torch

    import lurk_beta_amd as L
    from lurk_beta_amd import synth
    from lurk_beta_amd.spartan import SpartanProver

    log_n = min(args.log_n, 20)
    nc = nv = 1 << log_n
    nfree, nio = nv - nc // 2, 6  # half of the rows get a product variable; the rest of W is free
    F, q = L.FIELD_PALLAS_FQ, 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
    rng = np.random.default_rng(11)
    R = (1 << 256) % q
    tab = np.array([[(v * R % q) >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)] for v in (1, q - 1, 2, 3, 1 << 16, q - 7)], dtype=np.uint64)
    rows_p = nc // 2  # rows 0 .. rows_p - 1: (sum of free) * (sum of free) = product variable; the other rows are 0 * 0 = 0

    def rand_mat():
        cnt = np.zeros(nc, dtype=np.uint64)
        cnt[:rows_p] = rng.integers(2, 5, rows_p)
        indptr = np.zeros(nc + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        cols = rng.integers(0, nfree + 1 + nio, nnz)
        cols = np.where(cols >= nfree, cols - nfree + nv, cols).astype(np.uint64)  # free variables, then u and X behind the whole of W
        return indptr, cols, np.ascontiguousarray(tab[rng.integers(0, len(tab), nnz)])

    A, B = rand_mat(), rand_mat()
    cnt = np.zeros(nc, dtype=np.uint64)
    cnt[:rows_p] = 1
    ip = np.zeros(nc + 1, dtype=np.uint64)
    np.cumsum(cnt, out=ip[1:])
    Cm = (ip, (nfree + np.arange(rows_p)).astype(np.uint64), np.tile(tab[0], (rows_p, 1)))
    t0 = time.perf_counter()
    prover = SpartanProver(L.CURVE_PALLAS, q, [A, B, Cm], nc, nv, nio)
    setup_s = time.perf_counter() - t0
    # a strictly satisfying witness built on the device: free variables random, product variables = (A z)(B z) via the cross term of z with itself
    d_z = torch.zeros((nv + 1 + nio, 4), dtype=torch.int64, device="cuda")
    d_z[:nfree] = synth.scalars(F, 21, 1, nfree, mont=True)
    one = torch.from_numpy(tab[0:1].view(np.int64)).cuda()
    d_z[nv:nv + 1] = one
    d_z[nv + 1:] = synth.scalars(F, 22, 0, nio, mont=True)
    d_t = prover.shape.cross_term(d_z, d_z)                      # 2 (Az o Bz) - 2 u Cz, and Cz = 0 while the product variables are 0
    half = np.array([[((q + 1) // 2 * R % q) >> (64 * w) & 0xFFFFFFFFFFFFFFFF for w in range(4)]], dtype=np.uint64)
    d_prod = L.fold_vec(F, torch.zeros_like(d_t), d_t, half)      # (Az o Bz)
    d_z[nfree:nfree + rows_p] = d_prod[:rows_p]
    d_W = d_z[:nv].contiguous()
    d_E = torch.zeros((nc, 4), dtype=torch.int64, device="cuda")
    X = [int(v) for v in _ints_from(d_z[nv + 1:].cpu().numpy().view(np.uint64), R, q)]
    d_ck = synth.bases(L.CURVE_PALLAS, nc + 1)
    key = L.CommitmentKey(L.CURVE_PALLAS, d_ck, n=nc, device=True, precompute=bool(args.precompute))   # the prover's resident key (table by default)
    key.reserve(nc, 2)
    cw = key.commit_device(d_W, nv, is_mont=True)
    ce = key.commit_device(d_E, nc, is_mont=True)
    torch.cuda.synchronize()

    def step():
        return prover.prove(X, 1, d_W, d_E, d_ck, cw, ce, key=key if args.ipa_resident_key else None)

    for _ in range(args.warmup):
        step()
    lib.lurk_hip_profile_enable(1)
    lib.lurk_hip_profile_reset()
    torch.cuda.synchronize()
    # (as timeit does: no cyclic-garbage collection inside the timed region - a full collection is a 35 ms pause of the transcript callback)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    lib.lurk_hip_profile_enable(0)

    def kernel_ms(name):
        tot, cnt = ctypes.c_double(), ctypes.c_uint64()
        _lib.check(lib.lurk_hip_profile_get(name.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
        return round(tot.value / max(args.steps, 1), 3), cnt.value // max(args.steps, 1)

    from lurk_beta_amd import _lib

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        # the verifier's sum-check identity on the proof itself: the outer claim chain starts at 0 (a satisfied instance)
        p0 = proof["polys_outer"][0]
        assert (2 * p0[0] + sum(p0[1:])) % q == 0, "outer sum-check does not start from claim 0: the instance is not satisfied"
        verified = None
        if args.verify:
            # the oracle's VERIFIER (oracle/spartan_fast.py: the protocol of spartan_ref.py with the vector work in C) on the proof of the
            # last timed step, at the bench's own size; and on the same proof for a different statement, which it must reject
            from oracle import coracle as C
            from oracle import spartan_fast as SF

            t_v = time.perf_counter()
            mats_c = [(ip, ix, C.from_mont(1, d)) for ip, ix, d in (A, B, Cm)]
            ck_host = d_ck.cpu().numpy().view(np.uint64).reshape(-1, 8)
            aff = lambda j: (lambda a: None if a == (0, 0) else a)(L.point_to_affine(L.CURVE_PALLAS, j))
            accepted = SF.verify(0, mats_c, nc, nv, X, ck_host, aff(cw), aff(ce), 1, proof)
            rejected = not SF.verify(0, mats_c, nc, nv, [(X[0] + 1) % q] + X[1:], ck_host, aff(cw), aff(ce), 1, proof)
            if not (accepted and rejected):
                raise SystemExit(f"bench.py --verify: the oracle verifier {'rejects the proof' if not accepted else 'accepts the proof for another statement'}")
            verified = {"ok": True, "oracle_s": round(time.perf_counter() - t_v, 1),
                        "against": "oracle/spartan_fast.py verify(): accepts the proof of the last timed step, rejects it for X[0] + 1"}
        out = {"metric": "CompressedSNARK-style proofs/s (primary-curve Spartan prover stand-in, Pallas)", "value": round(1e3 / ms, 3), "unit": "proofs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u32x8 (255-bit Montgomery, integer VALU)", "data": "synthetic",
               "config": {"workload": f"Spartan-style proof of a satisfied relaxed R1CS instance, 2^{log_n} constraints x 2^{log_n} variables "
                                      f"({int(A[0][-1]) + int(B[0][-1]) + rows_p} non-zeros): 3 sum-checks ({log_n} + {log_n + 1} + {log_n} rounds), "
                                      f"transposed sparse mat-vec, inner-product argument over a 2^{log_n}-point key",
                          "note": "functional stand-in, not byte-compatible with arecibo; transcript and round glue in Python on the host",
                          "verified": verified,
                          "shape_setup_s_once": round(setup_s, 2)},
               "roofline": compress_roofline(lib, args, nc),
               "kernels_ms_per_proof": {k: kernel_ms(k) for k in ("sumcheck_round", "eq_evals", "r1cs_multiply_vec", "fold_vec", "ipa_inner_product",
                                                                   "ipa_fold_halves", "ipa_points_fold", "ipa_round_scalars", "ipa_coef_fold", "msm_accumulate", "msm_sort",
                                                                   "msm_reduce", "key_fold", "msm_precompute")},
               "ipa": ("rounds under the resident table key; the key folded ONCE after four rounds (lurk_hip_msm_ctx_fold_key_dev inside lurk_hip_ipa_prove_dev), the "
                       "other sixteen under the folded key" if os.environ.get("LURK_IPA_FOLD_MIN_LOG", "18") != "0" else
                       "rounds under the resident table key (LURK_IPA_FOLD_MIN_LOG=0: no key fold, the round-3 form)")
               if args.ipa_resident_key else "published form: key folded every round"}
        print(json.dumps(out), flush=True)
    key.close()
    prover.close()


def compress_roofline(lib, args, n):
    """The proof's dominant kernel is the bucket accumulation of its 40 + commitments (the opening argument's L and R of every round, under
    the resident key): mean launch from the HIP-event profile of the timed region, 96 B per point of the key."""
    from lurk_beta_amd import _lib

    tot, cnt = ctypes.c_double(), ctypes.c_uint64()
    _lib.check(lib.lurk_hip_profile_get(b"msm_accumulate", ctypes.byref(tot), ctypes.byref(cnt)))
    if not cnt.value:
        return None
    ms = tot.value / cnt.value
    b = 96.0 * n
    return {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": round(b / (ms * 1e-3) / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
            "frac": round(b / (ms * 1e-3) / 8e12, 6), "traffic": None, "avg_launch_ms": round(ms, 4), "launches_per_proof": cnt.value // max(args.steps, 1),
            "algorithmic_bytes_per_launch": b,
            "note": "32 B scalar + 64 B base per point of the 2^k-point key; in the opening argument half of every round's scalars are zero (composed scalars under "
                    "the resident key), so the launches are shorter than a dense commitment's; integer-VALU bound"}


def _ints_from(arr, R, q):
    Rinv = pow(R, q - 2, q)
    return [(int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192) * Rinv % q for r in arr.reshape(-1, 4)]


def store_hydrate_workload(args, lib, world, rank):
    """Store hydration (StoreCore::hydrate_z_cache, /root/reference/src/lem/store_core.rs:256-269) on the two DAG
    shapes that bound it: DEEP (a list of 400 distinct symbols: 25 wide levels of string / symbol hashing, then a spine of one cons per
    level) and WIDE (12 000 symbols under a balanced tree of conses: ~4 x 10^5 nodes, 22 levels).  A "step" is one whole hydration through
    lurk_hip_store_hydrate, host records in, host digests out (that IS the boundary: the store lives in host memory).  The CPU leg is the
    oracle's hydration (oracle.c: the same levels, every core) on the same DAGs; --verify compares every digest."""
    import numpy as np

    from lurk_beta_amd import store_hasher as SH
    from oracle import coracle as C

    F = 1
    out = None
    for name, dag in (("deep", SH.list_dag(400)), ("wide", SH.wide_dag(12000))):
        rec, vals = SH.encode(dag)
        hashed = int((rec[:, 0] != 0).sum())
        for _ in range(max(1, args.warmup)):
            got, levels = SH.hydrate_records(F, rec, vals)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            got, levels = SH.hydrate_records(F, rec, vals)
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        r = {"ms_per_hydration": round(ms, 3), "nodes": len(dag), "hashed_nodes": hashed, "levels": levels,
             "value": round(hashed / ms / 1e3, 4), "unit": "M hashed nodes/s"}
        want = None
        if not args.no_cpu_baseline or args.verify:
            C.store_hydrate(F, rec[:64], vals)  # constants + thread pool
            t1 = time.perf_counter()
            want, _ = C.store_hydrate(F, rec, vals)
            dt = time.perf_counter() - t1
            r["cpu_baseline"] = {"value": round(hashed / dt / 1e6, 4), "unit": "M hashed nodes/s", "ms": round(dt * 1e3, 2), "cores": C.lib().orc_num_threads(), "kind": "port",
                                 "sample": "the same DAG, whole, oracle/oracle.c: orc_store_hydrate (level by level, OpenMP inside a level, plain-schedule Poseidon)"}
            r["speedup_vs_cpu_leg"] = round(dt * 1e3 / ms, 2)
        if args.verify:
            assert np.array_equal(got, want), f"{name}: digests differ from the oracle"
            r["verified"] = True
        if name == "deep":
            out = {"metric": "store hydration throughput (deep list DAG; the wide DAG is the `wide` sub-record)", "value": r["value"], "unit": r["unit"], "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_hydration"], "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "u32x8 (255-bit Montgomery, integer VALU) + 4x64 host limbs for narrow levels", "data": "synthetic",
                   "config": {"workload": f"lurk_hip_store_hydrate, {len(dag)}-node list DAG ({levels} levels), host records in / host digests out",
                              "verified": r.get("verified")},
                   "roofline": {"bound": "hbm", "kernel": "poseidon_wide_kernel", "achieved": round(hashed * 5 * 32 / (ms * 1e-3) / 1e9, 4), "peak": 8000.0,
                                "unit": "GB/s", "frac": round(hashed * 5 * 32 / (ms * 1e-3) / 8e12, 8), "traffic": None,
                                "note": "bound by the DAG's depth (one Poseidon dependency chain per level: ~0.14 ms on a GPU lane, ~25-50 us on a host core), "
                                        "not by HBM or VALU throughput: levels of at most 6 nodes are hashed by the library's host Poseidon"},
                   "deep": r}
        else:
            out["wide"] = r
    if "cpu_baseline" in out["deep"]:
        out["cpu_baseline"] = out["deep"]["cpu_baseline"]
    if rank == 0:
        print(json.dumps(out), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher in the environment: re-run this command line under
    torch.distributed.run with one rank per GPU (rendezvous on 127.0.0.1); rank 0 of that job prints the JSON line."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def sub_records(args):
    """The default line's sub-records: child runs of this file (a fresh process each: its own HIP context, nothing shared with the
    timed region above), each with --verify, so that the driver's one command witnesses the folding step (BASELINE.json's first
    metric, through its synthetic stand-in), the 2^24 Poseidon tree (configs[2]), the 2^24 NTT and the compressing proof (f3) with
    their parity checks."""
    import subprocess

    common = ["--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--sub-records", "off", "--pmc", "off", "--verify"]
    legs = {
        "fold_step_rc100": ["--workload", "fold_step", "--rc", "100"],
        "poseidon_tree_2_24": ["--workload", "poseidon_tree", "--log-n", "24"],
        "ntt_2_24": ["--workload", "ntt", "--log-n", "24"],
        "compress_2_20": ["--workload", "compress", "--log-n", "20"],  # the compressing proof of a 2^20 x 2^20 instance; --verify = the oracle's verifier
    }
    if args.no_cpu_baseline:
        common.append("--no-cpu-baseline")
    out = {}
    for name, extra in legs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + common, capture_output=True, text=True, timeout=240)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
            else:
                out[name] = json.loads(lines[-1])
        except Exception as e:  # noqa: BLE001 (a sub-record must never cost the headline line)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


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


def collect_traffic(args):
    """HBM bytes per launch of msm_accumulate_kernel from the PMC counters, as MI355X_MICROARCH.md section HBM prescribes:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (they do not fit one), both in KiB, FETCH_SIZE doubled on
    gfx950 (128-byte requests tallied as 64).  Returns (bytes_per_launch or None, detail or None)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--pmc", "off", "--steps", "2", "--no-cpu-baseline",
             "--log-n", str(args.log_n), "--dist", args.dist, "--precompute", str(args.precompute), "--window-bits", str(args.window_bits)]
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    work = tempfile.mkdtemp(prefix="lurk_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            outdir = os.path.join(work, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "--"] + child
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            except Exception as e:  # noqa: BLE001
                return None, {"error": f"rocprofv3 --pmc {counter} failed: {type(e).__name__}"}
            per = []
            for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if "msm_accumulate_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                            per.append(float(r["Counter_Value"]))
            if not per:
                return None, {"error": f"no {counter} rows for msm_accumulate_kernel"}
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
    import glob as _glob
    for pth in sorted(_glob.glob(os.path.join(ROOT, "profiles", "*fetch_calibration.json")), reverse=True):
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
                                 "the guide's x2 is for wide streaming reads only and is listed beside it; WRITE_SIZE as is"}


def poseidon_mads_per_hash(arity):
    """v_mad_u64_u32 per hash of the kernels' schedule (poseidon29.cuh), radix-2^29 layer: product 135, squaring 99, one lazy row
    of k terms 81 k + 54.  Full round: t S-boxes (2 squarings + 1 product) + t rows of t terms; partial round: 1 S-box + one
    row of t terms + t - 1 products; canonical in (arity products) and out (1)."""
    from oracle import pyref as R

    t = arity + 1
    rf, rp = R.round_numbers(arity)
    sbox, row = 2 * 99 + 135, 81 * t + 54
    return rf * (t * sbox + t * row) + rp * (sbox + row + (t - 1) * 135) + (arity + 1) * 135


def poseidon_valu_roofline(arity, hashes, kernel_ms):
    mads = poseidon_mads_per_hash(arity)
    peak = 1024 * 2.15e9 * 64 / (mads * 4.6)  # hashes / s if the SIMDs issued nothing but those mads (4.6 cycles per wave-instruction, measured)
    ach = hashes / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    return {"bound": "valu", "kernel": "poseidon_batch_kernel", "achieved": round(ach / 1e6, 2), "peak": round(peak / 1e6, 2), "unit": f"M hash{arity}/s",
            "frac": round(ach / peak, 4), "mads_per_hash": mads}


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


def other_workloads(args, lib, world, rank):
    """Poseidon arity-8 tree (BASELINE configs[2]: 2^24 Pallas-Fq leaves) and the radix-2 NTT, same timing
    contract: inputs resident in HBM, K timed steps.  N > 1: the tree is ONE tree sharded by subtrees with a
    single 8 x 32-byte all-gather (SURVEY.md section 8e, strong scaling); the NTT runs replicas."""
    import numpy as np
    import torch
    import torch.distributed as dist

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
            out["cpu_baseline"] = {"value": round(m / dt / 1e6, 4), "unit": unit, "cores": C.lib().orc_num_threads(), "kind": "port",
                                   "sample": f"{sample_desc}, {dt:.2f} s (oracle/oracle.c, OpenMP)"}
        print(json.dumps(out), flush=True)


def cpu_quota_cores():
    """CPUs the cgroup lets this container use (cpu.max / cfs quota), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            q = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            per = float(fh.read())
        return None if q <= 0 else q / per
    except Exception:
        return None


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


if __name__ == "__main__":
    main()

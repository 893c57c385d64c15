#!/usr/bin/env python3
"""bench.py - Pedersen MSM throughput of the HIP hot path (BASELINE.json metric: "MSM Mscalar-mul/s").

A "step" is one commitment  C = sum_i s_i * ck_i  over Pallas: scalars and the commitment key are
already resident in HBM when the timed region starts (the PCIe-inclusive rate is noted in
DESIGN.md, never here).  By default four commitments are in flight (`--pipeline 4`, four of a context's async slots: two
accumulate, one is in its tail, one sorts; alternating runs on one box, K = 20: 3 / 4 / 5 / 6 in flight = 994 / 1 010 / 910 / 952
Mscalar-mul/s): every step is still one complete MSM and all K results
are produced inside the timed region, which starts and ends with an empty pipeline (so a small K pays the fill and
drain: K = 20 / 40 / 80 measure about 4.15 / 4.11 / 4.07 ms per step); `--pipeline 1` is the fully synchronous form, and
the default line also carries `sync_ms_per_commit`, the plain-key synchronous sub-record and the one-shot
host-pointer sub-record.  The resident key
carries the precomputed per-window table by default (`--precompute 0` = plain 64 B/point key).  N = 1: n = 2^log_n points on one GPU (default 2^22, the size the metric
is quoted at; `--log-n 20` is BASELINE.json configs[1]).  N > 1: one process per GPU, every rank
owns its own 2^log_n-point shard of an (N * 2^log_n)-point commitment (weak scaling); each step
ends with the path's real exchange: an RCCL all_gather of the 96-byte partial commitments and the
group sum on every rank.

The workloads live in bench_workloads/ (msm = this file's default and the driver's line; fold_step, compress, poseidon_tree, ntt,
store_hydrate = the other kernels of the path, also carried by the default line as verified sub-records whose compact summary is the
LAST key of the line).  The K-step region is repeated `--reps` times and the median is reported.  Exit status: non-zero when a parity
check (`--verify`, `cpu_baseline.matches_gpu_result`) or a sub-record fails.

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
    ap.add_argument("--reps", type=int, default=5,
                    help="msm: repetitions of the --steps region (each bracketed by barrier + synchronize); the MEDIAN region is what the line reports")
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--dist", choices=["uniform", "witness"], default="uniform")
    ap.add_argument("--precompute", type=int, default=1,
                    help="1 = resident key with the per-window precomputed table (13 x 64 B per point, 20-bit windows); 0 = plain key")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="commitments in flight (1 = synchronous; 2..6 = async slots: the tail of one overlaps the accumulation of the next).  4 since round 5: "
                         "two accumulate, one is in its tail, one sorts (profiles/r05_pipeline_depth.txt: 3 / 4 / 5 / 6 in flight = 994 / 1 010 / 910 / 952)")
    ap.add_argument("--stage-ahead", type=int, default=None,
                    help="fold_step: 3 (default since round 6) = the witness producer runs TWO steps ahead: witness k+2 is traced from step k's submit hook, instance k+1 "
                         "(complete) is staged when step k starts and step k's begin submits its commitment behind commit(T) in the FOLLOW class "
                         "(LURK_MSM_SUBMIT_FOLLOW: low-priority sort + plan at once, persistent accumulation once commit(T)'s has ended); "
                         "1 = traced and staged one step ahead at the top of the step; 2 = the same from inside begin (the submit hook); "
                         "0 = plain begin, W2 of step k committed inside step k (rounds 1-5; the default with --devices, where nothing is staged ahead)")
    ap.add_argument("--witness-ahead", type=int, default=3,
                    help="fold_step: the witness of step k+1 is traced while step k is in progress (the reference's producer thread): 3 = enqueued from the step's submit hook "
                         "(lurk_hip_fold_ctx_set_submit_hook: behind the step's opening kernels, beside its commitments; default: 3.5 ms at rc = 100); 2 = enqueued after begin "
                         "has returned, so that it runs while the host derives r (3.85); 1 = enqueued ahead of this step's commitments (4.2); 0 = traced at the start of its own step")
    ap.add_argument("--secondary", type=int, default=1, help="fold_step: 1 = also time the secondary-curve (Vesta, ~10^4 constraints) half of a step")
    ap.add_argument("--late-ranges", type=int, default=1, help="fold_step with --stage-ahead: 1 = 12 000 positions of W2 arrive with begin (the augmented circuit's), 0 = none")
    ap.add_argument("--ipa-resident-key", type=int, default=1, help="compress: 1 = inner-product rounds under the resident key (composed scalars), 0 = fold the key")
    ap.add_argument("--spartan-prover", choices=["library", "python"], default="library",
                    help="compress: library = lurk_hip_spartan_prove_dev, the prover as one call (default); python = the same entry points sequenced by lurk_beta_amd/spartan.py")
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
    ap.add_argument("--auto-slices", type=int, default=0,
                    help="fold_step --devices: 1 = LURK_MSM_FLAG_AUTO_SLICES (use only as many of the listed devices as leave every slice >= 2^20 points)")
    ap.add_argument("--helper-devices", default="",
                    help="fold_step with --stage-ahead 1: comma-separated devices that each hold a copy of the commitment key and commit the instances staged "
                         "ahead in turn (lurk_hip_fold_ctx_add_helper: staging ahead across GPUs); e.g. 1,2,3 on a node, 0 on a one-GPU box (functional)")
    ap.add_argument("--shape-file", default="", help="fold_step: a LURKDUMP R1CS shape (lurk_beta_amd/dump.py; written by rust/lurk-hip-sys/src/dump.rs from arecibo's "
                                                     "R1CSShape) instead of the synthetic step circuit; needs --witness-file")
    ap.add_argument("--witness-file", default="", help="fold_step: LURKDUMP witnesses (W, X of consecutive steps + pp_digest) for --shape-file")
    ap.add_argument("--key-file", default="", help="fold_step with --shape-file: LURKDUMP commitment key (default: the synthetic key of the same length)")
    ap.add_argument("--sub-records", choices=["auto", "off"], default="auto",
                    help="auto = the default msm line at N = 1 also carries the other workloads of the path as verified sub-records "
                         "(fold_step_rc100, poseidon_tree_2_24, ntt_2_24, compress_2_20: each a child run of this file with --verify, same --steps / --warmup)")
    args = ap.parse_args()
    if args.stage_ahead is None:
        args.stage_ahead = 0 if (args.devices or args.shape_file) else 3
    if args.workload == "fold_step":
        # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues PER PRIORITY CLASS (default 4), balancing them by use count; the step
        # keeps ~25 streams busy (commitment slots, the folding contexts of both curves, the witness producer) and two streams that
        # share a queue serialise.  8 per class is the measured best for the step (round 6, both curves alive, staged flow: 4 / 6 / 8 / 12
        # = 3.33 / 3.32 / 3.18 / 3.32 ms, and at 12 the queues of one process exceed what the device keeps mapped and it time-slices
        # them: 17 ms with one more producer).  The library helps by not opening a third class: a slot's low-priority accumulate
        # stream is only made when a commitment needs it (msm.hip: ensure_acc_stream) - with it open, 8 already over-subscribed (8.9 ms).
        # The msm workload does not care (1 016 / 1 011 Mscalar-mul/s at 4 / 8).  Must be set before the HIP runtime initialises.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # N > 1 without a launcher: become the launcher (one rank per GPU, the same command line the driver uses)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from bench_workloads.sub_records import spawn_ranks

        return spawn_ranks(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or drop WORLD_SIZE")

    import torch
    import torch.distributed as dist

    from lurk_beta_amd import _lib

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

    if args.workload == "fold_step" and (args.shape_file or args.witness_file):
        if not (args.shape_file and args.witness_file):
            sys.exit("bench.py: --shape-file and --witness-file go together")
        from bench_workloads.fold_step_files import fold_step_from_files

        return fold_step_from_files(args, lib, world, rank)
    if args.workload == "fold_step":
        from bench_workloads.fold_step import fold_step_workload

        return fold_step_workload(args, lib, world, rank)
    if args.workload == "compress":
        from bench_workloads.compress import compress_workload

        return compress_workload(args, lib, world, rank)
    if args.workload == "store_hydrate":
        from bench_workloads.store_hydrate import store_hydrate_workload

        return store_hydrate_workload(args, lib, world, rank)
    if args.workload != "msm":
        from bench_workloads.kernels import other_workloads

        return other_workloads(args, lib, world, rank)
    from bench_workloads.msm import msm_workload

    return msm_workload(args, lib, world, rank)


if __name__ == "__main__":
    sys.exit(main() or 0)

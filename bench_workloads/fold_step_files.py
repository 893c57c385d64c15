"""--workload fold_step --shape-file S --witness-file W [--key-file K]: one Nova folding step over DUMPED inputs - arecibo's own
R1CSShape, the fresh witnesses of consecutive steps and (optionally) the commitment key, written by rust/lurk-hip-sys/src/dump.rs
(format: lurk_beta_amd/dump.py) from /root/reference/benches/fibonacci.rs:98-122 - instead of the synthetic model of fold_step.py.
Same step entry points, same timing contract, same oracle check (--verify): what INTEGRATION.md section "Measuring the real fibonacci
step" runs the day a Rust host exists."""
from __future__ import annotations

import json
import time

from .common import kernel_profile


def fold_step_from_files(args, lib, world, rank):
    import numpy as np
    import torch

    import lurk_beta_amd as L
    from lurk_beta_amd import dump, synth

    sh = dump.read_shape(args.shape_file)
    wt = dump.read_witnesses(args.witness_file)
    F = sh["field_id"]
    assert F in (L.FIELD_PALLAS_FP, L.FIELD_PALLAS_FQ), "the folding step runs over the Pasta cycle (field id 0 or 1)"
    assert wt["field_id"] == F and wt["num_vars"] == sh["num_vars"] and wt["num_io"] == sh["num_io"], "witness file and shape file disagree"
    curve = L.CURVE_PALLAS if F == L.FIELD_PALLAS_FQ else L.CURVE_VESTA  # the curve whose scalar field the shape is over
    base_field = L.FIELD_PALLAS_FP if curve == L.CURVE_PALLAS else L.FIELD_PALLAS_FQ
    nc, nv, nio = sh["num_cons"], sh["num_vars"], sh["num_io"]
    n_key = max(nc, nv)
    stream = torch.cuda.current_stream().cuda_stream

    def mont_host(field, arr, encoding):  # canonical dumps go through the library's own fold kernel once (setup, not timed)
        if encoding == dump.ENC_MONTGOMERY or arr.size == 0:
            return np.ascontiguousarray(arr)
        d = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).cuda()
        return dump.to_montgomery_device(field, d).cpu().numpy().view(np.uint64)

    t_setup = time.perf_counter()
    mats = [(ip, ix, mont_host(F, d, sh["encoding"])) for ip, ix, d in sh["mats"]]
    shape = L.R1CSShape(F, nc, nv, nio, *mats)
    shape_setup_s = time.perf_counter() - t_setup
    if args.key_file:
        kf = dump.read_key(args.key_file)
        assert kf["curve"] == curve and kf["points"].shape[0] >= n_key, "key file: another curve, or fewer points than max(num_cons, num_vars)"
        pts = kf["points"][:n_key]
        if kf["encoding"] == dump.ENC_CANONICAL:
            pts = mont_host(base_field, pts.reshape(-1, 4), dump.ENC_CANONICAL).reshape(-1, 8)
        d_bases = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda()
        key_src = f"key file ({kf['points'].shape[0]} points)"
    else:  # a commitment's cost does not depend on which points the key holds: the bench's deterministic synthetic key
        d_bases = synth.bases(curve, n_key)
        key_src = "synthetic key (no --key-file)"
    ck = L.CommitmentKey(curve, d_bases, n=n_key, device=True, precompute=bool(args.precompute), window_bits=args.window_bits)
    ck.reserve(n_key, 4)
    ctx = L.FoldingContext(curve, shape, ck)  # the running pair starts as the default relaxed instance (all zero), as RecursiveSNARK::new's does
    ctx.set_pp_digest(wt["pp_digest"])
    steps = []
    for w, x in wt["steps"]:
        w_m = mont_host(F, w, wt["encoding"])
        x_m = mont_host(F, x, wt["encoding"]) if nio else np.zeros((0, 4), dtype=np.uint64)
        steps.append((torch.from_numpy(w_m.view(np.int64)).cuda(), np.ascontiguousarray(x_m)))
    torch.cuda.synchronize()
    k = [0]

    def step():
        d_w2, x2 = steps[k[0] % len(steps)]
        k[0] += 1
        cw, ct = ctx.begin(d_w2, x2, stream=stream)
        r = ctx.challenge()
        ctx.finish(r)
        return cw, ct

    for _ in range(args.warmup):
        step()
    lib.lurk_hip_profile_enable(1)
    lib.lurk_hip_profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.lurk_hip_profile_enable(0)
    verified = None
    if args.verify and rank == 0:
        assert curve == L.CURVE_PALLAS, "--verify of a dumped step: the oracle leg is written for the primary (Pallas) curve"
        from .fold_step import verify_fold_step

        q = dump._MODULUS[F]
        d_w2, x2 = steps[k[0] % len(steps)]
        buf = torch.empty_like(d_w2)
        verified = verify_fold_step(L, ctx, mats, F, q, nv, nc, nio, d_bases, wt["pp_digest"], x2, lambda b: b.copy_(d_w2), buf)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        info = shape.info()
        acc_ms, acc_n = kernel_profile(lib, "msm_accumulate")
        ct_ms, ct_n = kernel_profile(lib, "r1cs_cross_term")
        acc_avg = acc_ms / max(acc_n, 1)
        acc_bytes = 96.0 * (nv + nc) / 2.0
        frames = max(1, round(nc / 10973))  # Lurk's step circuit has 10 973 constraints per frame on the Pasta fields (fold_step.py)
        out = {"metric": "Lurk iterations/s (one Nova folding step over dumped inputs, primary curve)", "value": round(frames / (ms * 1e-3), 1),
               "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (255-bit Montgomery, integer VALU)", "data": "dumped",
               "config": {"workload": f"fold step over {args.shape_file} ({nc} constraints x {nv} variables, {sum(info['nnz'])} non-zeros, {info['distinct_coefficients']} distinct "
                                      f"coefficients) and {len(steps)} dumped witness(es) of {args.witness_file}; {key_src}",
                          "frames_per_step_assumed": frames, "verified": verified, "shape_setup_s_once": round(shape_setup_s, 2),
                          "note": "W2 resident in HBM when begin is called (the PCIe copy of a host-synthesized witness is 32 B x num_vars on top); "
                                  "transcript derived per step by the library (lurk_hip_fold_step_challenge)"},
               "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": round(acc_bytes / (acc_avg * 1e-3) / 1e9, 3) if acc_avg else None, "peak": 8000.0,
                            "unit": "GB/s", "frac": round(acc_bytes / (acc_avg * 1e-3) / 8e12, 6) if acc_avg else None, "traffic": None,
                            "avg_launch_ms": round(acc_avg, 4), "algorithmic_bytes_per_launch": acc_bytes},
               "fold_kernels": {"r1cs_cross_term_ms": round(ct_ms / max(ct_n, 1), 4)}}
        print(json.dumps(out), flush=True)
    ctx.close()
    ck.close()
    shape.close()
    return 0

"""--workload compress: the compressing (Spartan-style) prover at the rc = 100 step circuit's padded size."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time

from .common import BENCH, ROOT


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

    in_library = bool(args.ipa_resident_key) and args.spartan_prover == "library"  # the prover as ONE library call (what a Rust caller binds)

    def step():
        return prover.prove(X, 1, d_W, d_E, d_ck, cw, ce, key=key if args.ipa_resident_key else None, in_library=in_library)

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
                          "note": "functional stand-in, not byte-compatible with arecibo; " +
                                  ("the whole prover is ONE library call (lurk_hip_spartan_prove_dev: transcript, round loops and scratch arena inside)" if in_library else
                                   "the prover's sequence driven from Python (lurk_beta_amd/spartan.py), round loops and transcript in the library"),
                          "prover": "library" if in_library else "python",
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

"""Host-side mirror of arecibo's ``SumcheckProof::prove_cubic_with_additive_term`` / ``prove_quad`` over the HIP library
(SURVEY.md section 8 f3): what ``CompressedSNARK::prove`` (/root/reference/src/proof/nova.rs:341-356) spends its time in between
its opening MSMs.  The tables live in HBM (torch tensors of Montgomery elements), one library call per round binds the previous
challenge and sums the next round's evaluations; the transcript is a callback, as the Keccak transcript stays on the host."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def _ints(a: np.ndarray) -> list[int]:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a]


def _limbs(vals) -> np.ndarray:
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for w in range(4):
            out[i, w] = (int(v) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def eq_evals(field_id: int, r_mont: np.ndarray, stream=None):
    """``EqPolynomial::new(r).evals()`` as a device tensor of 2^len(r) Montgomery elements."""
    import torch

    r = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(-1, 4)
    out = torch.empty((1 << r.shape[0], 4), dtype=torch.int64, device="cuda")
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.load().lurk_hip_eq_evals_dev(field_id, _lib.ptr(r), r.shape[0], _lib.ptr(out), _lib.ptr(s)))
    return out


def prove(field_id: int, modulus: int, claim: int, tables, challenge, stream=None):
    """tables: 4 device tensors (A, B, C, D: comb = A (B C - D)) or 2 (A, B: comb = A B), Montgomery, length 2^k; they are
    consumed (bound in place).  challenge(round, poly_coeffs) -> r (int).  All integers here are canonical; the Montgomery
    factor R = 2^256 is stripped from / applied to what crosses the library boundary.
    Returns (round polynomials, final evaluations P_k(r), final claim) like the oracle's sumcheck_prove."""
    import torch

    lib = _lib.load()
    p = modulus
    R = (1 << 256) % p
    Rinv = pow(R, p - 2, p)
    cubic = len(tables) == 4
    assert len(tables) in (2, 4)
    n = tables[0].shape[0]
    assert all(t.is_cuda and t.shape[0] == n for t in tables) and n >= 2 and n & (n - 1) == 0
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    ptrs = (ctypes.c_void_p * len(tables))(*[_lib.ptr(t) for t in tables])
    nv = 3 if cubic else 2
    inv2, inv6 = pow(2, p - 2, p), pow(6, p - 2, p)
    polys, r_prev, length = [], None, n
    rounds = n.bit_length() - 1
    for j in range(rounds):
        ev = np.zeros((nv, 4), dtype=np.uint64)
        if r_prev is None:
            _lib.check(lib.lurk_hip_sumcheck_round_dev(field_id, 3 if cubic else 2, ptrs, length, None, _lib.ptr(ev), _lib.ptr(s)))
        else:
            rm = _limbs([r_prev * R % p])
            _lib.check(lib.lurk_hip_sumcheck_round_dev(field_id, 3 if cubic else 2, ptrs, length, _lib.ptr(rm), _lib.ptr(ev), _lib.ptr(s)))
            length //= 2
        # comb multiplies 2 (quadratic) or 3 (cubic: a * b * c, and a * d) Montgomery factors; the library's products divide by R
        # once each, so a sum of a*b carries R^1 like any Montgomery value: strip one R
        e = [x * Rinv % p for x in _ints(ev)]
        e0, e2 = e[0], e[1]
        e1 = (claim - e0) % p
        if cubic:
            e3 = e[2]
            d = e0
            a3 = (e3 - 3 * e2 + 3 * e1 - e0) * inv6 % p
            b = ((e2 - 2 * e1 + e0) * inv2 - 3 * a3) % p
            c = (e1 - d - a3 - b) % p
            poly = [d, c, b, a3]
        else:
            a2 = (e2 - 2 * e1 + e0) * inv2 % p
            poly = [e0, (e1 - e0 - a2) % p, a2]
        polys.append(poly)
        r_prev = int(challenge(j, poly)) % p
        acc = 0
        for co in reversed(poly):
            acc = (acc * r_prev + co) % p
        claim = acc
    rm = _limbs([r_prev * R % p])
    _lib.check(lib.lurk_hip_sumcheck_round_dev(field_id, 3 if cubic else 2, ptrs, length, _lib.ptr(rm), None, _lib.ptr(s)))
    torch.cuda.synchronize()
    finals = [_ints(t[:1].cpu().numpy().view(np.uint64))[0] * Rinv % p for t in tables]
    return polys, finals, claim

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
    consumed (bound in place).  challenge(round, poly_coeffs) -> r (int), or a ``_lib.KeccakRounds`` (the library's own transcript: its
    ``challenges()`` then holds the r's); all integers here are canonical.  The round loop is host
    code of the library (lurk_hip_sumcheck_prove_dev: what a Rust caller binds), the transcript calls back into ``challenge``.
    Returns (round polynomials, final evaluations P_k(r), final claim) like the oracle's sumcheck_prove."""
    import torch

    lib = _lib.load()
    p = modulus
    cubic = len(tables) == 4
    assert len(tables) in (2, 4)
    n = tables[0].shape[0]
    assert all(t.is_cuda and t.shape[0] == n for t in tables) and n >= 2 and n & (n - 1) == 0
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    ptrs = (ctypes.c_void_p * len(tables))(*[_lib.ptr(t) for t in tables])
    rounds, ncoef = n.bit_length() - 1, 4 if cubic else 3
    out_polys = np.zeros((rounds, ncoef, 4), dtype=np.uint64)
    out_finals = np.zeros((len(tables), 4), dtype=np.uint64)
    out_claim = np.zeros(4, dtype=np.uint64)
    failure = []

    def on_round(_user, j, coef_ptr, out_ptr):
        try:
            co = np.ctypeslib.as_array(ctypes.cast(coef_ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(ncoef * 4,)).copy()
            r = int(challenge(j, _ints(co))) % p
            ctypes.memmove(out_ptr, r.to_bytes(32, "little"), 32)
            return 0
        except BaseException as e:  # noqa: BLE001 - an exception must not unwind through the C frames
            failure.append(e)
            return 1

    if isinstance(challenge, _lib.KeccakRounds):  # the transcript is the library's own: no Python in the round loop
        challenge.struct.n_scalars = ncoef
        cb_ptr, user = challenge.callback("sumcheck")
    else:
        cb = _lib.SUMCHECK_CHALLENGE_FN(on_round)
        cb_ptr, user = ctypes.cast(cb, ctypes.c_void_p), None
    rc = lib.lurk_hip_sumcheck_prove_dev(field_id, 3 if cubic else 2, ptrs, n, _lib.ptr(_limbs([claim % p])), cb_ptr, user,
                                         _lib.ptr(out_polys), _lib.ptr(out_finals), _lib.ptr(out_claim), _lib.ptr(s))
    if failure:
        raise failure[0]
    _lib.check(rc)
    return [_ints(out_polys[j]) for j in range(rounds)], _ints(out_finals), _ints(out_claim)[0]


def _prove_batch(field_id: int, modulus: int, degree: int, groups, coeffs, claim: int, challenge, stream=None):
    """The batched round loop is host code of the library (lurk_hip_sumcheck_prove_batch_dev: what a Rust caller binds); this wrapper
    only marshals.  groups: per instance its 2 (degree 2) or 4 (degree 3) device tables, all of one length (consumed)."""
    import torch

    lib = _lib.load()
    p = modulus
    np_t = 4 if degree == 3 else 2
    n = groups[0][0].shape[0]
    assert all(len(g) == np_t and all(t.is_cuda and t.shape[0] == n for t in g) for g in groups) and n >= 2 and n & (n - 1) == 0
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    ptrs = (ctypes.c_void_p * (np_t * len(groups)))(*[_lib.ptr(t) for g in groups for t in g])
    rounds, ncoef = n.bit_length() - 1, degree + 1
    out_polys = np.zeros((rounds, ncoef, 4), dtype=np.uint64)
    out_finals = np.zeros((len(groups) * np_t, 4), dtype=np.uint64)
    out_claim = np.zeros(4, dtype=np.uint64)
    rs, failure = [], []

    def on_round(_user, j, coef_ptr, out_ptr):
        try:
            co = np.ctypeslib.as_array(ctypes.cast(coef_ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(ncoef * 4,)).copy()
            r = int(challenge(_ints(co))) % p
            rs.append(r)
            ctypes.memmove(out_ptr, r.to_bytes(32, "little"), 32)
            return 0
        except BaseException as e:  # noqa: BLE001 - an exception must not unwind through the C frames
            failure.append(e)
            return 1

    if isinstance(challenge, _lib.KeccakRounds):
        challenge.struct.n_scalars = ncoef
        first = int(challenge.struct.n_rounds)
        cb_ptr, user = challenge.callback("sumcheck")
    else:
        cb = _lib.SUMCHECK_CHALLENGE_FN(on_round)
        cb_ptr, user = ctypes.cast(cb, ctypes.c_void_p), None
    rc = lib.lurk_hip_sumcheck_prove_batch_dev(field_id, degree, len(groups), ptrs, n, _lib.ptr(_limbs([c % p for c in coeffs])), _lib.ptr(_limbs([claim % p])),
                                               cb_ptr, user, _lib.ptr(out_polys), _lib.ptr(out_finals), _lib.ptr(out_claim), _lib.ptr(s))
    if failure:
        raise failure[0]
    _lib.check(rc)
    if isinstance(challenge, _lib.KeccakRounds):
        rs = challenge.challenges()[first:]
    fin = _ints(out_finals)
    return [_ints(out_polys[j]) for j in range(rounds)], rs, [tuple(fin[i * np_t:(i + 1) * np_t]) for i in range(len(groups))], _ints(out_claim)[0]


def prove_quad_batch(field_id: int, modulus: int, claims, pairs, coeffs, challenge, stream=None):
    """sum_i coeff_i * sum_x A_i(x) B_i(x) with ONE challenge per round shared by every pair (the evaluation-claim batching of
    arecibo's snark.rs).  pairs: [(A_i, B_i)] device tensors of one common length (consumed); challenge(poly) -> r.
    Returns (round polynomials, challenges, [(A_i(r), B_i(r))], final claim)."""
    claim = sum(c * e for c, e in zip(coeffs, claims)) % modulus
    return _prove_batch(field_id, modulus, 2, [tuple(pr) for pr in pairs], coeffs, claim, challenge, stream)


def prove_cubic_batch(field_id: int, modulus: int, quads, coeffs, challenge, stream=None):
    """sum_i coeff_i * sum_x A_i(x) (B_i(x) C_i(x) - D_i(x)) with ONE challenge per round shared by every instance: the outer sum-check
    of the batched SNARK (arecibo's spartan::batched; /root/reference/src/proof/supernova.rs:110).  quads: [(A_i, B_i, C_i, D_i)] device
    tensors of one common length (consumed; a table may be shared by several instances only if it is passed as separate copies).  The
    claim starts at 0 (every instance satisfied).  Returns (round polynomials, challenges, [final (A, B, C, D)(r) per instance], claim)."""
    return _prove_batch(field_id, modulus, 3, [tuple(q) for q in quads], coeffs, 0, challenge, stream)

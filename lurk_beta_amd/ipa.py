"""Host-side mirror of arecibo's ``ipa_pc::InnerProductArgument::prove`` over the HIP library: the
opening argument of CompressedSNARK on the Pasta cycle (/root/reference/src/proof/nova.rs:57-62, 341-356).  The vectors and the
commitment key stay in HBM; per round the library computes the cross inner products, the two commitments and, with the
transcript's challenge, the folds.  The transcript is a callback.  Two forms: with the prover's resident ``CommitmentKey``
(``key=``) the key is never folded - L and R are commitments under the original key of the composed scalar vectors, one pair
commitment per round - and the round loop itself is host code of the library (lurk_hip_ipa_prove_dev: what a Rust caller binds;
48 ms per 2^20 proof against 54 with the loop in Python); without it the published form (MSM over the halves of the folded device
key, lurk_hip_points_fold_halves_dev), whose loop is below."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .msm import point_sum
from .step import point_mul


def _limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)


def _int(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(4)
    return int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192


def _prove_resident_key(curve: int, q: int, key, ck_c: np.ndarray, d_a, d_b, challenge, s):
    """The rounds under the ORIGINAL resident key (see the module docstring): the loop itself is host code of the library
    (lurk_hip_ipa_prove_dev); the transcript calls back into ``challenge``.  d_a, d_b are folded in place."""
    lib = _lib.load()
    n0 = d_a.shape[0]
    rounds = n0.bit_length() - 1
    Ls = np.zeros((max(rounds, 1), 12), dtype=np.uint64)
    Rs = np.zeros((max(rounds, 1), 12), dtype=np.uint64)
    a_hat = np.zeros(4, dtype=np.uint64)
    ck_hat = np.zeros(8, dtype=np.uint64)
    failure = []

    def on_round(_user, j, l_ptr, r_ptr, out_ptr):
        try:
            L = np.ctypeslib.as_array(ctypes.cast(l_ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(12,)).copy()
            Rr = np.ctypeslib.as_array(ctypes.cast(r_ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(12,)).copy()
            r = int(challenge(j, L, Rr)) % q
            ctypes.memmove(out_ptr, r.to_bytes(32, "little"), 32)
            return 0
        except BaseException as e:  # noqa: BLE001 - an exception must not unwind through the C frames
            failure.append(e)
            return 1

    if isinstance(challenge, _lib.KeccakRounds):  # the transcript is the library's own: no Python in the round loop
        cb_ptr, user = challenge.callback("ipa")
    else:
        cb = _lib.IPA_CHALLENGE_FN(on_round)
        cb_ptr, user = ctypes.cast(cb, ctypes.c_void_p), None
    rc = lib.lurk_hip_ipa_prove_dev(key._ctx, _lib.ptr(d_a), _lib.ptr(d_b), n0, _lib.ptr(ck_c), cb_ptr, user, _lib.ptr(Ls), _lib.ptr(Rs),
                                    _lib.ptr(a_hat), _lib.ptr(ck_hat), _lib.ptr(s))
    if failure:
        raise failure[0]
    _lib.check(rc)
    return [Ls[j].copy() for j in range(rounds)], [Rs[j].copy() for j in range(rounds)], _int(a_hat), ck_hat


def prove(curve: int, order: int, d_ck, ck_c_jac: np.ndarray, d_a, d_b, r0: int, challenge, stream=None, key=None):
    """key: the prover's resident ``CommitmentKey`` over the same n bases (then d_ck may be None and the key is never folded).
    d_ck: (n, 8) int64 device tensor of affine Montgomery points (consumed: folded in place of a scratch copy); ck_c_jac: the
    extra base as a 96-byte Jacobian; d_a, d_b: (n, 4) Montgomery scalars on the device (consumed).  r0: the transcript's first
    challenge (scales ck_c); challenge(round, L, R) -> r.  Returns (L_vec, R_vec, a_hat) with points as 96-byte Jacobians and
    a_hat canonical."""
    import torch

    lib = _lib.load()
    sf = 1 if curve == 0 else 0  # scalar field id
    q = order
    R = (1 << 256) % q
    Rinv = pow(R, q - 2, q)
    mont = lambda v: _limbs(v * R % q)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    n = d_a.shape[0]
    assert n >= 1 and n & (n - 1) == 0 and d_b.shape[0] == n
    r0m = mont(r0)
    ck_c = point_mul(curve, ck_c_jac, r0m)
    if key is not None:
        return _prove_resident_key(curve, q, key, ck_c, d_a, d_b, challenge, s)
    assert d_ck.shape[0] == n
    ctx = ctypes.c_void_p()
    _lib.check(lib.lurk_hip_msm_ctx_create_dev(ctypes.byref(ctx), curve, _lib.ptr(d_ck), n, 0, _lib.ptr(s)))
    ck_cur, ck_next = d_ck, torch.empty((max(n // 2, 1), 8), dtype=torch.int64, device=d_ck.device)
    Ls, Rs = [], []
    j = 0
    try:
        while n > 1:
            h = n // 2
            cl, cr = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
            _lib.check(lib.lurk_hip_inner_product_dev(sf, _lib.ptr(d_a), _lib.ptr(d_b[h:]), h, _lib.ptr(cl), _lib.ptr(s)))
            _lib.check(lib.lurk_hip_inner_product_dev(sf, _lib.ptr(d_a[h:]), _lib.ptr(d_b), h, _lib.ptr(cr), _lib.ptr(s)))
            parts = []
            for bases, scal, c in ((ck_cur[h:], d_a, cl), (ck_cur, d_a[h:], cr)):
                _lib.check(lib.lurk_hip_msm_ctx_rebind_dev(ctx, _lib.ptr(bases), h))
                m = np.zeros(12, dtype=np.uint64)
                _lib.check(lib.lurk_hip_msm_ctx_run_dev(ctx, _lib.ptr(m), _lib.ptr(scal), h, 1, _lib.ptr(s)))
                parts.append(point_sum(curve, np.stack([m, point_mul(curve, ck_c, c)])))
            L, Rr = parts
            Ls.append(L)
            Rs.append(Rr)
            r = challenge.ipa_round(j, L, Rr) if isinstance(challenge, _lib.KeccakRounds) else int(challenge(j, L, Rr)) % q
            ri = pow(r, q - 2, q)
            rm, rim = mont(r), mont(ri)  # named: the arrays must outlive the calls that read them
            _lib.check(lib.lurk_hip_fold_halves_dev(sf, _lib.ptr(d_a), n, _lib.ptr(rm), _lib.ptr(rim), _lib.ptr(s)))
            _lib.check(lib.lurk_hip_fold_halves_dev(sf, _lib.ptr(d_b), n, _lib.ptr(rim), _lib.ptr(rm), _lib.ptr(s)))
            _lib.check(lib.lurk_hip_points_fold_halves_dev(curve, _lib.ptr(ck_cur), n, _lib.ptr(rim), _lib.ptr(rm), _lib.ptr(ck_next), _lib.ptr(s)))
            ck_cur, ck_next = ck_next, ck_cur
            n = h
            j += 1
        torch.cuda.synchronize()
        a_hat = _int(d_a[:1].cpu().numpy().view(np.uint64)) * Rinv % q
        return Ls, Rs, a_hat, ck_cur[:1].cpu().numpy().view(np.uint64).reshape(8)
    finally:
        lib.lurk_hip_msm_ctx_destroy(ctx)

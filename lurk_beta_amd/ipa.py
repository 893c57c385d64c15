"""Host-side mirror of arecibo's ``ipa_pc::InnerProductArgument::prove`` over the HIP library: the
opening argument of CompressedSNARK on the Pasta cycle (/root/reference/src/proof/nova.rs:57-62, 341-356).  The vectors and the
commitment key stay in HBM; per round the library computes the cross inner products, the two commitments and, with the
transcript's challenge, the folds.  The transcript is a callback.  Two forms: with the prover's resident ``CommitmentKey``
(``key=``) the key is never folded - L and R are commitments under the original key of the composed scalar vectors
(lurk_hip_ipa_round_scalars_dev), two table-mode MSMs in flight per round; without it the published form (MSM over the halves of
the folded device key, lurk_hip_points_fold_halves_dev)."""
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


_BASE_MODULUS = {0: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,   # Pallas: coordinates in Fp
                 1: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001}   # Vesta: coordinates in Fq


def _prove_resident_key(curve: int, q: int, key, ck_c: np.ndarray, d_a, d_b, challenge, s):
    """The rounds under the ORIGINAL resident key (see the module docstring); d_a, d_b are folded in place."""
    import torch

    from .msm import point_to_affine

    lib = _lib.load()
    sf = 1 if curve == 0 else 0
    R = (1 << 256) % q
    Rinv = pow(R, q - 2, q)
    mont = lambda v: _limbs(v * R % q)
    n0 = d_a.shape[0]
    d_coef = torch.from_numpy(np.tile(mont(1), (n0, 1)).view(np.int64)).to(d_a.device)
    pairs = key.supports_pairs()  # a window-table key commits L and R (disjoint supports) as two key spaces of ONE pass
    d_l = torch.empty_like(d_coef)
    d_r = None if pairs else torch.empty_like(d_coef)
    Ls, Rs = [], []
    m, j = n0, 0
    while m > 1:
        h = m // 2
        cl, cr = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        _lib.check(lib.lurk_hip_ipa_round_scalars_dev(sf, _lib.ptr(d_a), m, _lib.ptr(d_coef), n0, _lib.ptr(d_l), _lib.ptr(d_r), _lib.ptr(s)))
        if pairs:
            key.submit_pair_device(0, d_l, n0, h.bit_length() - 1, is_mont=True, stream=s)   # bit log2(m / 2) of the index: set = L's support
        else:
            key.submit_device(0, d_l, n0, is_mont=True, stream=s, mode=1)   # both commitments in flight under the inner products; the host
            key.submit_device(1, d_r, n0, is_mont=True, stream=s, mode=1)   # waits for both: the foreground class (plain accumulate launch)
        _lib.check(lib.lurk_hip_inner_product_dev(sf, _lib.ptr(d_a), _lib.ptr(d_b[h:]), h, _lib.ptr(cl), _lib.ptr(s)))
        _lib.check(lib.lurk_hip_inner_product_dev(sf, _lib.ptr(d_a[h:]), _lib.ptr(d_b), h, _lib.ptr(cr), _lib.ptr(s)))
        tl, trr = point_mul(curve, ck_c, cl), point_mul(curve, ck_c, cr)  # two 255-bit host scalar multiples (0.2 ms each): under the MSM
        if pairs:
            c_r, c_l = key.wait_pair(0)
        else:
            c_l, c_r = key.wait(0), key.wait(1)
        L = point_sum(curve, np.stack([c_l, tl]))
        Rr = point_sum(curve, np.stack([c_r, trr]))
        Ls.append(L)
        Rs.append(Rr)
        r = int(challenge(j, L, Rr)) % q
        ri = pow(r, q - 2, q)
        rm, rim = mont(r), mont(ri)  # named: the arrays must outlive the calls that read them
        _lib.check(lib.lurk_hip_fold_halves_dev(sf, _lib.ptr(d_a), m, _lib.ptr(rm), _lib.ptr(rim), _lib.ptr(s)))
        _lib.check(lib.lurk_hip_fold_halves_dev(sf, _lib.ptr(d_b), m, _lib.ptr(rim), _lib.ptr(rm), _lib.ptr(s)))
        _lib.check(lib.lurk_hip_ipa_coef_fold_dev(sf, _lib.ptr(d_coef), n0, m, _lib.ptr(rim), _lib.ptr(rm), _lib.ptr(s)))
        m = h
        j += 1
    # the final key element is the commitment of the coefficient vector (the verifier's s vector)
    x, y = point_to_affine(curve, key.commit_device(d_coef, n0, is_mont=True, stream=s))
    p = _BASE_MODULUS[curve]
    Rb = (1 << 256) % p
    ck_hat = np.concatenate([_limbs(x * Rb % p), _limbs(y * Rb % p)]) if (x, y) != (0, 0) else np.zeros(8, dtype=np.uint64)
    torch.cuda.synchronize()
    a_hat = _int(d_a[:1].cpu().numpy().view(np.uint64)) * Rinv % q
    return Ls, Rs, a_hat, ck_hat


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
            r = int(challenge(j, L, Rr)) % q
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

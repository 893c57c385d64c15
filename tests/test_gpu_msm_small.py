"""The small-commitment path (lurk_beta_amd/csrc/msm_small.hip): a resident key of <= 2^16 points created with the precompute
flag keeps every multiple of every window base, and a commitment is one launch summing gathered points.  Bit-exact against the
oracle (naive double-and-add, oracle Pippenger, msm_fast.c) for n in 1 .. 2^16 on both curves, with the edge-case list of
test_gpu_msm.py.  Callers it serves: the secondary-curve commitments of every folding step
(/root/reference/src/proof/nova.rs:291-293) and SuperNova's small circuits (/root/reference/src/proof/supernova.rs:242-244)."""
import ctypes

import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu
CURVES = [("pallas", 0), ("vesta", 1)]
_sf = lambda c: 1 - c


def _info(key):
    from lurk_beta_amd import _lib

    cv, n, wb, pre = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.load().lurk_hip_msm_ctx_info(key._ctx, ctypes.byref(cv), ctypes.byref(n), ctypes.byref(wb), ctypes.byref(pre)))
    assert pre.value in (0, 1)  # a boolean; the form comes from lurk_hip_msm_ctx_form
    _lib.check(_lib.load().lurk_hip_msm_ctx_form(key._ctx, ctypes.byref(pre)))
    return cv.value, n.value, wb.value, pre.value


@pytest.mark.parametrize("cn,c", CURVES)
def test_sizes_one_to_2_16(hip, cn, c):
    from lurk_beta_amd import CommitmentKey, point_to_affine

    sf = _sf(c)
    for n in (1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1000, 4096, 8191, 10000, 1 << 14, (1 << 14) + 1, 40000, 1 << 16):
        B = C.synth_bases(c, n)
        key = CommitmentKey(c, B, precompute=True)
        assert _info(key)[2:] == (8 if n <= (1 << 14) else 6, 2), n  # form 2 = the small form: 8-bit windows up to 2^14 points, 6-bit above
        for dist in (0, 1):
            S = C.synth_scalars(sf, 11 + dist, dist, n)
            want = C.jac_to_affine(c, C.msm_naive(c, B, S) if n <= 256 else C.msm_fast(c, B, S))
            assert point_to_affine(c, key.commit(S)) == want, (n, dist)
            assert point_to_affine(c, key.commit(C.to_mont(sf, S), is_mont=True)) == want, (n, dist, "mont")
        if n > 3:  # CE::commit(ck, v) uses ck[..v.len()]
            m = n // 3
            S = C.synth_scalars(sf, 13, 0, m)
            assert point_to_affine(c, key.commit(S)) == C.jac_to_affine(c, C.msm_fast(c, B[:m], S) if m > 256 else C.msm_naive(c, B[:m], S)), (n, m)
        assert point_to_affine(c, key.commit(np.zeros((0, 4), dtype=np.uint64))) == (0, 0)
        key.close()


@pytest.mark.parametrize("cn,c", CURVES)
def test_edge_cases(hip, cn, c):
    from lurk_beta_amd import CommitmentKey, point_to_affine

    q = R.CURVES[cn]["order"]
    n = 64
    B = C.synth_bases(c, n)
    B[1] = 0                      # identity base (0,0): every multiple is the identity record
    B[3] = B[2]                   # repeated base: equal partial sums meet in the butterfly -> the doubling branch
    B[5] = B[4]
    s = [R.uniform_fe(9, i, q) for i in range(n)]
    s[2] = s[3] = 12345
    s[4], s[5] = 777, q - 777     # P and -P: an identity in the middle of the tree
    s[6] = 0
    s[7] = 1
    s[8] = q - 1
    s[9] = 0x80                   # digit exactly 2^7 (the largest multiple of the 8-bit table)
    s[10] = 0x81                  # first value that recodes to a negative digit with carry
    s[11] = (1 << 254) | 0xFFFF   # carries rippling from the bottom, top window in use
    s[12] = int("ff" * 31, 16) % q
    s[13] = int("7f" * 31, 16)
    s[14] = int("80" * 31, 16)
    S = C.ints_to_limbs(s)
    want = C.jac_to_affine(c, C.msm_naive(c, B, S))
    key = CommitmentKey(c, B, precompute=True)
    assert point_to_affine(c, key.commit(S)) == want
    out = key.commit(np.zeros((n, 4), dtype=np.uint64))   # all-zero scalars -> identity, z = 0
    assert point_to_affine(c, out) == (0, 0) and not out[8:].any()
    key.close()
    # every scalar identical, every base identical with scalar 1 (n P through doublings), P - P everywhere
    n2 = 3000
    B2 = C.synth_bases(c, n2)
    k2 = CommitmentKey(c, B2, precompute=True)
    S2 = np.tile(C.ints_to_limbs([s[0]]), (n2, 1))
    assert point_to_affine(c, k2.commit(S2)) == C.jac_to_affine(c, C.msm_fast(c, B2, S2))
    k2.close()
    B3 = np.tile(B[:1], (300, 1))
    k3 = CommitmentKey(c, B3, precompute=True)
    assert point_to_affine(c, k3.commit(C.ints_to_limbs([1] * 300))) == C.jac_to_affine(c, C.msm_naive(c, B3, C.ints_to_limbs([1] * 300)))
    alt = C.ints_to_limbs([5 if i % 2 == 0 else q - 5 for i in range(300)])
    assert point_to_affine(c, k3.commit(alt)) == (0, 0)
    k3.close()
    # 6-bit table edge digits
    n4 = (1 << 14) + 8
    B4 = C.synth_bases(c, n4)
    k4 = CommitmentKey(c, B4, precompute=True)
    e4 = [0, 1, q - 1, 0x20, 0x21, 0x1F, (1 << 254) | 0x3F, int("3f" * 31, 16) % q] + [R.uniform_fe(10, i, q) for i in range(n4 - 8)]
    S4 = C.ints_to_limbs(e4)
    assert point_to_affine(c, k4.commit(S4)) == C.jac_to_affine(c, C.msm_fast(c, B4, S4))
    k4.close()


@pytest.mark.parametrize("cn,c", CURVES)
def test_slots_in_flight_and_against_the_bucket_path(hip, cn, c):
    """Four commitments in flight on the key's slots (device-resident scalars); the same commitments through the bucket pipeline
    (window-bit override: the round-2 form of a small table key) and through a plain key must agree bit for bit."""
    import torch

    from lurk_beta_amd import CommitmentKey, point_to_affine

    sf, n = _sf(c), 10000
    B = C.synth_bases(c, n)
    small = CommitmentKey(c, B, precompute=True)
    bucket = CommitmentKey(c, B, precompute=True, window_bits=16)
    plain = CommitmentKey(c, B)
    assert _info(small)[2] == 8 and _info(bucket)[2] == 16
    small.reserve(n, 4)
    vecs = [C.synth_scalars(sf, 20 + k, k % 2, n if k != 2 else 777) for k in range(4)]
    dev = [torch.from_numpy(C.to_mont(sf, v).view(np.int64)).cuda() for v in vecs]
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(4):
            small.submit_device(k, dev[k], len(vecs[k]), is_mont=True)
        for k in (2, 0, 3, 1):
            got = point_to_affine(c, small.wait(k))
            assert got == point_to_affine(c, bucket.commit(vecs[k])) == point_to_affine(c, plain.commit(vecs[k])), (rep, k)
            assert got == C.jac_to_affine(c, C.msm_fast(c, B[: len(vecs[k])], vecs[k]))
    for k in (small, bucket, plain):
        k.close()


def test_key_file_round_trip(hip, tmp_path):
    from lurk_beta_amd import CommitmentKey, point_to_affine

    c, n = 0, 5000
    B = C.synth_bases(c, n)
    key = CommitmentKey(c, B, precompute=True)
    S = C.synth_scalars(1, 30, 1, n)
    want = point_to_affine(c, key.commit(S))
    path = str(tmp_path / "small.key")
    key.save(path, with_table=True)   # the small form's table is not written: 64 B per point, rebuilt on load
    import os

    assert os.path.getsize(path) == 64 + n * 64
    k2 = CommitmentKey.load(path, precompute=True)
    assert _info(k2)[2:] == (8, 2)  # form 2: the small form again
    assert point_to_affine(c, k2.commit(S)) == want == C.jac_to_affine(c, C.msm_fast(c, B, S))
    k3 = CommitmentKey.load(path, precompute=False)
    assert point_to_affine(c, k3.commit(S)) == want
    for k in (key, k2, k3):
        k.close()

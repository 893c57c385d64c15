"""f3 (opening half): the rounds of the inner-product argument on the device against the oracle's restatement of the published
argument (oracle/pyref.py: ipa_prove / ipa_verify), on both curves; the device-assisted proof must also pass the oracle's
VERIFIER (the size-independent property).  Parity unpinned upstream."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


_ORACLE = {}


def _oracle_case(cn, c, log_n):
    """Inputs and the oracle's proof, computed once per (curve, size): the three forms of the device prover share them."""
    if (cn, log_n) in _ORACLE:
        return _ORACLE[(cn, log_n)]
    sf = 1 if c == 0 else 0
    q = R.CURVES[cn]["order"]
    n = 1 << log_n
    B = C.synth_bases(c, n + 1)
    bf = 0 if c == 0 else 1  # base field id
    ck_pts = [tuple(C.limbs_to_ints(C.from_mont(bf, B[i].reshape(2, 4)))) for i in range(n + 1)]
    ck, ck_c = ck_pts[:n], ck_pts[n]
    a = C.limbs_to_ints(C.synth_scalars(sf, 160, 1, n))
    b = C.limbs_to_ints(C.synth_scalars(sf, 161, 0, n))
    r0 = R.uniform_fe(162, 0, q)
    chal = [R.uniform_fe(163, j, q) for j in range(log_n)]
    want = R.ipa_prove(cn, ck, ck_c, a, b, r0, chal)
    comm_a = R.msm_naive(cn, a, ck)
    _ORACLE[(cn, log_n)] = (B, ck, ck_c, a, b, r0, chal, want, comm_a)
    return _ORACLE[(cn, log_n)]


def _aff(curve_id, jac):
    from lurk_beta_amd import point_to_affine

    xy = point_to_affine(curve_id, jac)
    return None if xy == (0, 0) else xy


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
@pytest.mark.parametrize("log_n", [1, 4, 7, 9])
@pytest.mark.parametrize("form", ["folded_key", "resident_plain_key", "resident_table_key", "resident_window_table_key", "window_table_key_folded_after_4_rounds",
                                  "window_table_key_folded_after_3_rounds", "window_table_key_folded_after_2_rounds"])
def test_ipa_rounds_match_the_oracle_and_verify(hip, cn, c, log_n, form, monkeypatch):
    """resident_table_key: the small-commitment form at these sizes (two commitments per round); resident_window_table_key: the bucket
    pipeline with the window table (L and R as one pair commitment), the key never folded; ..._folded_after_4_rounds: the same with the
    key folded once by the first four rounds' weights (lurk_hip_msm_ctx_fold_key_dev inside lurk_hip_ipa_prove_dev, threshold moved down
    to these sizes; at 2^9 the folded key of 32 points is folded a second time) - all five forms must give the oracle's proof."""
    from lurk_beta_amd import CommitmentKey, ipa, msm

    if log_n == 9 and form in ("folded_key", "resident_plain_key"):
        pytest.skip("covered at the smaller sizes")
    monkeypatch.setenv("LURK_IPA_FOLD_MIN_LOG", "5" if "_folded_after_" in form else "0")
    if "_folded_after_" in form:  # LURK_IPA_FOLD_ROUNDS: rounds under the long key before the fold (the default is 4)
        monkeypatch.setenv("LURK_IPA_FOLD_ROUNDS", form.split("_folded_after_")[1][0])
        if log_n < 7 and form[-8] != "4":
            pytest.skip("covered by the 4-round form at these sizes")

    sf = 1 if c == 0 else 0
    bf = 0 if c == 0 else 1
    q = R.CURVES[cn]["order"]
    n = 1 << log_n
    B, ck, ck_c, a, b, r0, chal, (want_L, want_R, want_a, want_ck), comm_a = _oracle_case(cn, c, log_n)
    ck_c_jac = np.concatenate([B[n], C.to_mont(bf, C.ints_to_limbs([1])).reshape(4)])
    # the resident key may be longer than the argument (Spartan opens under a prefix of the witness key)
    key = None if form == "folded_key" else CommitmentKey(c, B[: n + 1], precompute=form != "resident_plain_key",
                                                          window_bits=16 if "window_table" in form else 0)
    got_L, got_R, got_a, got_ck = ipa.prove(c, q, None if key else _dev(B[:n]), ck_c_jac, _dev(C.to_mont(sf, C.ints_to_limbs(a))),
                                            _dev(C.to_mont(sf, C.ints_to_limbs(b))), r0, lambda j, L, Rr: chal[j], key=key)
    if key:
        key.close()
    assert [_aff(c, x) for x in got_L] == want_L and [_aff(c, x) for x in got_R] == want_R
    assert got_a == want_a
    assert tuple(C.limbs_to_ints(C.from_mont(bf, got_ck.reshape(2, 4)))) == want_ck
    if form == "folded_key":  # (the other forms produced the identical proof: one run of the oracle's verifier covers them)
        cc = sum(x * y for x, y in zip(a, b)) % q
        assert R.ipa_verify(cn, ck, ck_c, comm_a, b, cc, r0, chal, [_aff(c, x) for x in got_L], [_aff(c, x) for x in got_R], got_a)
        assert not R.ipa_verify(cn, ck, ck_c, comm_a, b, (cc + 1) % q, r0, chal, want_L, want_R, want_a)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("log_n,log_t,window_bits", [(10, 4, 16), (10, 0, 16), (12, 8, 16), (17, 4, 0), (17, 1, 0), (12, 5, 20)])
def test_key_fold_against_the_oracle(hip, curve, log_n, log_t, window_bits):
    """lurk_hip_msm_ctx_fold_key_dev: out[p] = sum_b w_b * key[b m + p] against the oracle's Pippenger on sampled outputs (first, last,
    random): weights 0, 1, q - 1, 2^254 and uniform ones; identity points and a repeated point among the bases (mixed additions that
    double or cancel inside a bucket); a key longer than the folded prefix; one window group and several (m U < 2^17 lanes)."""
    from lurk_beta_amd import CommitmentKey

    sf = 1 if curve == 0 else 0
    bf = 0 if curve == 0 else 1
    q = R.modulus(sf)
    n, T = 1 << log_n, 1 << log_t
    m = n // T
    B = C.synth_bases(curve, n + 3).copy()
    B[5] = 0                      # identity points
    B[n - 1] = 0
    if T > 1:
        B[m + 7] = B[7]           # the same point in two blocks of output 7: equal weights double it, opposite weights cancel it
    key = CommitmentKey(curve, B, precompute=True, window_bits=window_bits)
    assert key.info()["form"] == "table"
    special = [0, 1, q - 1, 1 << 254]
    w = [special[b] if b < len(special) else R.uniform_fe(180 + curve, b, q) for b in range(T)]
    if T >= 2:
        w[1] = (q - w[0]) % q if w[0] else w[1]
    if T >= 4:
        w[3] = w[2]
    got = key.fold_key(n, C.to_mont(sf, C.ints_to_limbs(w))).cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(4)
    sample = sorted({0, 5, 7, m - 1, *[int(x) for x in rng.integers(0, m, 12)]} & set(range(m)))
    wl = C.ints_to_limbs(w)
    for p in sample:
        want = C.jac_to_affine(curve, C.msm_pippenger(curve, np.ascontiguousarray(B[p:n:m]), wl))
        g = tuple(C.limbs_to_ints(C.from_mont(bf, got[p].reshape(2, 4))))
        assert g == want, (p, log_n, log_t)
    key.close()


def test_fold_kernels_edge_cases(hip):
    """Identity points, equal halves (L = R: the joint ladder's L + R is a doubling), opposite halves (L + R = identity), zero and
    one scalars."""
    import torch

    from lurk_beta_amd import _lib

    lib = _lib.load()
    c, cn, bf, sf = 0, "pallas", 0, 1
    q = R.CURVES[cn]["order"]
    B = C.synth_bases(c, 8)
    pts = [tuple(C.limbs_to_ints(C.from_mont(bf, B[i].reshape(2, 4)))) for i in range(8)]
    half = [pts[0], None, pts[2], pts[3]] + [pts[0], pts[1], R.ec_neg(cn, pts[2]), None]
    arr = np.zeros((8, 8), dtype=np.uint64)
    for i, P in enumerate(half):
        if P is not None:
            arr[i] = C.to_mont(bf, C.ints_to_limbs(list(P))).reshape(8)
    R256 = (1 << 256) % q
    for lo, hi in ((R.uniform_fe(170, 0, q), R.uniform_fe(170, 1, q)), (1, 1), (0, 5), (7, 0), (q - 1, 1)):
        out = torch.zeros((4, 8), dtype=torch.int64, device="cuda")
        d_arr, lo_m, hi_m = _dev(arr), C.ints_to_limbs([lo * R256 % q]), C.ints_to_limbs([hi * R256 % q])
        _lib.check(lib.lurk_hip_points_fold_halves_dev(c, _lib.ptr(d_arr), 8, _lib.ptr(lo_m), _lib.ptr(hi_m), _lib.ptr(out), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        for i in range(4):
            want = R.ec_add(cn, R.ec_mul(cn, lo, half[i]) if half[i] else None, R.ec_mul(cn, hi, half[4 + i]) if half[4 + i] else None)
            g = tuple(C.limbs_to_ints(C.from_mont(bf, got[i].reshape(2, 4))))
            assert (None if g == (0, 0) else g) == want, (lo, hi, i)


def test_library_round_loop_contract(hip):
    """lurk_hip_ipa_prove_dev (the round loop as host code of the library): a one-element argument has no rounds, an exception of the
    transcript callback aborts the call and comes back as itself, a challenge that is not reduced below the group order or is zero is
    refused, and so are a length that is no power of two and a key shorter than the vectors."""
    import ctypes

    import torch

    from lurk_beta_amd import CommitmentKey, LurkHipError, _lib, ipa

    lib = _lib.load()
    c, cn, sf, bf = 0, "pallas", 1, 0
    q = R.CURVES[cn]["order"]
    B = C.synth_bases(c, 9)
    key = CommitmentKey(c, B[:8], precompute=False)
    ck_c_jac = np.concatenate([B[8], C.to_mont(bf, C.ints_to_limbs([1])).reshape(4)])
    a = C.limbs_to_ints(C.synth_scalars(sf, 180, 1, 8))
    b = C.limbs_to_ints(C.synth_scalars(sf, 181, 0, 8))
    vec = lambda v: _dev(C.to_mont(sf, C.ints_to_limbs(v)))
    # n = 1: no rounds, a_hat = a[0], the key element = ck[0]
    L, Rr, a_hat, ck_hat = ipa.prove(c, q, None, ck_c_jac, vec(a[:1]), vec(b[:1]), 5, lambda j, l, r: 1 / 0, key=key)
    assert L == [] and Rr == [] and a_hat == a[0]
    assert tuple(C.limbs_to_ints(C.from_mont(bf, ck_hat.reshape(2, 4)))) == tuple(C.limbs_to_ints(C.from_mont(bf, B[0].reshape(2, 4))))

    class Boom(Exception):
        pass

    def bad(j, l, r):
        raise Boom("transcript")

    with pytest.raises(Boom):
        ipa.prove(c, q, None, ck_c_jac, vec(a), vec(b), 5, bad, key=key)
    # the slot the aborted call had submitted on must be usable again
    want = R.ipa_prove(cn, [tuple(C.limbs_to_ints(C.from_mont(bf, B[i].reshape(2, 4)))) for i in range(8)],
                       tuple(C.limbs_to_ints(C.from_mont(bf, B[8].reshape(2, 4)))), a, b, 5, [3, 4, 6])
    got = ipa.prove(c, q, None, ck_c_jac, vec(a), vec(b), 5, lambda j, l, r: [3, 4, 6][j], key=key)
    assert [_aff(c, x) for x in got[0]] == want[0] and got[2] == want[2]

    # raw C ABI: unreduced and zero challenges, bad lengths
    out = [np.zeros((3, 12), dtype=np.uint64), np.zeros((3, 12), dtype=np.uint64), np.zeros(4, dtype=np.uint64), np.zeros(8, dtype=np.uint64)]

    def call(n, r_value, d_a=None, d_b=None):
        def cb(_u, j, l, r, o):
            ctypes.memmove(o, int(r_value).to_bytes(32, "little"), 32)
            return 0

        fn = _lib.IPA_CHALLENGE_FN(cb)
        from lurk_beta_amd.step import point_mul

        ck_c = point_mul(c, ck_c_jac, C.to_mont(sf, C.ints_to_limbs([5])).reshape(4))
        return lib.lurk_hip_ipa_prove_dev(key._ctx, _lib.ptr(d_a if d_a is not None else vec(a)), _lib.ptr(d_b if d_b is not None else vec(b)), n,
                                          _lib.ptr(ck_c), ctypes.cast(fn, ctypes.c_void_p), None, _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]),
                                          _lib.ptr(out[3]), None)

    assert call(8, q) != 0 and b"reduced" in lib.lurk_hip_last_error()
    assert call(8, 0) != 0 and b"zero" in lib.lurk_hip_last_error()
    assert call(6, 3) != 0 and b"power of two" in lib.lurk_hip_last_error()
    big = torch.zeros((16, 4), dtype=torch.int64, device="cuda")
    assert call(16, 3, big, big) != 0 and b"fewer points" in lib.lurk_hip_last_error()
    assert call(8, 3) == 0
    with pytest.raises(LurkHipError):
        _lib.check(call(8, q))
    key.close()

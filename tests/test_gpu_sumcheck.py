"""f3 (partial): the sum-check rounds of CompressedSNARK::prove on the device against the oracle's restatement of the published
Spartan prover (oracle/pyref.py: sumcheck_prove), plus the verifier's own checks as size-independent properties at 2^20.
Parity unpinned upstream (no proof bytes exist; arecibo is un-vendored)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


@pytest.mark.parametrize("f", [0, 1, 2])
@pytest.mark.parametrize("ell,tail", [(1, None), (2, None), (5, None), (11, None), (1, "0"), (5, "0"), (11, "0"), (5, "3"), (11, "11"), (11, "1")])
def test_rounds_match_the_oracle(hip, f, ell, tail, monkeypatch):
    """tail: LURK_SUMCHECK_HOST_TAIL_LOG - the rounds move to the host once the tables are down to 2^tail elements (default 2^8; "0":
    every round on the device; 2^11 here: every round on the host) - the proof must not depend on where that is."""
    from lurk_beta_amd import sumcheck as S

    if tail is not None:
        monkeypatch.setenv("LURK_SUMCHECK_HOST_TAIL_LOG", tail)
    p = R.modulus(f)
    n = 1 << ell
    for ntab in (4, 2):
        tabs = [C.synth_scalars(f, 120 + k + ntab, k % 2, n) for k in range(ntab)]
        ints = [C.limbs_to_ints(t) for t in tabs]
        comb = (lambda a, b, c, d: a * (b * c - d)) if ntab == 4 else (lambda a, b: a * b)
        claim = sum(comb(*[t[i] for t in ints]) for i in range(n)) % p
        chal = [R.uniform_fe(130 + ntab, j, p) for j in range(ell)]
        want_polys, want_finals, want_claim = R.sumcheck_prove(p, claim, ints, chal)
        got_polys, got_finals, got_claim = S.prove(f, p, claim, [_dev(C.to_mont(f, t)) for t in tabs], lambda j, poly: chal[j])
        assert got_polys == want_polys and got_finals == want_finals and got_claim == want_claim, (f, ell, ntab)
        assert got_claim == comb(*got_finals) % p  # the verifier's final check


@pytest.mark.parametrize("f", [0, 1])
def test_eq_evals(hip, f):
    from lurk_beta_amd import sumcheck as S

    p = R.modulus(f)
    for ell in (0, 1, 4, 9):
        r = [R.uniform_fe(140, j, p) for j in range(ell)]
        got = C.limbs_to_ints(C.from_mont(f, S.eq_evals(f, C.to_mont(f, C.ints_to_limbs(r)) if ell else np.zeros((0, 4), dtype=np.uint64)).cpu().numpy().view(np.uint64)))
        assert got == R.eq_evals(p, r), (f, ell)
        assert sum(got) % p == 1  # the eq polynomial's evaluations sum to one


def test_outer_sumcheck_at_2_20_satisfies_the_verifier(hip):
    """The outer sum-check of a satisfied relaxed instance at 2^20 rows, tables generated in HBM: claim 0 = sum_x eq(tau, x) (Az Bz - (u Cz + E))
    with E := Az Bz - u Cz (fold kernels).  Checks what a verifier checks - p_j(0) + p_j(1) = claim_j for every round and the final
    claim = eq(tau, r) (Az(r) Bz(r) - D(r)) - which is independent of the size."""
    import torch

    from lurk_beta_amd import _lib, synth
    from lurk_beta_amd import sumcheck as S

    f, ell = 1, 20
    p = R.modulus(f)
    n = 1 << ell
    tau = [R.uniform_fe(150, j, p) for j in range(ell)]
    A = S.eq_evals(f, C.to_mont(f, C.ints_to_limbs(tau)))
    B = synth.scalars(f, 151, 1, n, mont=True)
    Cc = synth.scalars(f, 152, 0, n, mont=True)
    # D := B o C (a satisfied instance: the summand vanishes row by row) through a one-table quadratic round is not needed: build D on the host for 2^12 rows
    # and check the identity on the device tables by the sum-check itself: with D = B o C the claim is 0.
    Bh, Ch = C.from_mont(f, B.cpu().numpy().view(np.uint64)), C.from_mont(f, Cc.cpu().numpy().view(np.uint64))
    D = _dev(C.to_mont(f, C.mul_canonical(f, Bh, Ch)))
    chal = [R.uniform_fe(153, j, p) for j in range(ell)]
    claims = [0]

    def challenge(j, poly):
        assert (2 * poly[0] + sum(poly[1:])) % p == claims[-1] % p  # p(0) + p(1) = claim
        claims.append(R.unipoly_eval(p, poly, chal[j]))
        return chal[j]

    polys, finals, claim = S.prove(f, p, 0, [A, B, Cc, D], challenge)
    assert len(polys) == ell and claim == claims[-1]
    assert claim == finals[0] * (finals[1] * finals[2] - finals[3]) % p
    # eq(tau, r) evaluated directly
    eq_r = 1
    for t, r in zip(tau, chal):
        eq_r = eq_r * ((t * r + (1 - t) * (1 - r)) % p) % p
    assert finals[0] == eq_r


def test_bad_arguments(hip):
    import ctypes

    import torch

    from lurk_beta_amd import LurkHipError, _lib

    t = torch.zeros((6, 4), dtype=torch.int64, device="cuda")
    ptrs = (ctypes.c_void_p * 2)(_lib.ptr(t), _lib.ptr(t))
    ev = np.zeros((3, 4), dtype=np.uint64)
    with pytest.raises(LurkHipError):
        _lib.check(_lib.load().lurk_hip_sumcheck_round_dev(1, 2, ptrs, 6, None, _lib.ptr(ev), None))  # not a power of two
    with pytest.raises(LurkHipError):
        _lib.check(_lib.load().lurk_hip_sumcheck_round_dev(1, 5, ptrs, 4, None, _lib.ptr(ev), None))  # unknown degree


def test_library_round_loop_contract(hip):
    """lurk_hip_sumcheck_prove_dev: an exception of the transcript callback aborts the call and comes back as itself; a challenge or a
    claim that is not reduced below the field order is refused; so is a table length that is no power of two."""
    import ctypes

    from lurk_beta_amd import _lib
    from lurk_beta_amd import sumcheck as S

    lib = _lib.load()
    f = 1
    p = R.modulus(f)
    tabs = lambda: [_dev(C.to_mont(f, C.synth_scalars(f, 150 + k, 0, 8))) for k in range(2)]

    class Boom(Exception):
        pass

    def bad(j, poly):
        raise Boom("transcript")

    with pytest.raises(Boom):
        S.prove(f, p, 5, tabs(), bad)

    def call(n, r_value, claim=5):
        t = tabs()
        ptrs = (ctypes.c_void_p * 2)(*[_lib.ptr(x) for x in t])

        def cb(_u, j, co, o):
            ctypes.memmove(o, int(r_value).to_bytes(32, "little"), 32)
            return 0

        fn = _lib.SUMCHECK_CHALLENGE_FN(cb)
        out = [np.zeros((3, 3, 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64)]
        claim_l = np.array([(claim >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)
        return lib.lurk_hip_sumcheck_prove_dev(f, 2, ptrs, n, _lib.ptr(claim_l), ctypes.cast(fn, ctypes.c_void_p), None, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                               _lib.ptr(out[2]), None)

    assert call(8, p) != 0 and b"challenge is not reduced" in lib.lurk_hip_last_error()
    assert call(8, 3, claim=p) != 0 and b"claim is not reduced" in lib.lurk_hip_last_error()
    assert call(6, 3) != 0 and b"power of two" in lib.lurk_hip_last_error()
    assert call(8, 3) == 0

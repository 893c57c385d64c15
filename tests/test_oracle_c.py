"""The fast C oracle (oracle/oracle.c) against the KAT-pinned Python restatement.  CPU only."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R


@pytest.mark.parametrize("f", [0, 1, 2])
def test_fields_and_generators(f):
    p = R.modulus(f)
    a = [R.uniform_fe(7, i, p) for i in range(50)] + [0, 1, p - 1]
    A = C.ints_to_limbs(a)
    assert C.limbs_to_ints(C.from_mont(f, C.to_mont(f, A))) == a
    assert C.limbs_to_ints(C.to_mont(f, A)) == [x * (1 << 256) % p for x in a]
    assert C.limbs_to_ints(C.synth_scalars(f, 7, 0, 50)) == a[:50]
    w = [R.witness_like_fe(1, i, p) for i in range(400)]
    assert C.limbs_to_ints(C.synth_scalars(f, 1, 1, 400)) == w
    assert C.dot(f, A, A) == sum(x * x for x in a) % p


@pytest.mark.parametrize("f", [0, 1, 2])
@pytest.mark.parametrize("arity", [3, 4, 6, 8])
def test_poseidon_matches_pyref(f, arity):
    p = R.modulus(f)
    pre = [[R.uniform_fe(2, i * arity + j, p) for j in range(arity)] for i in range(4)]
    pre.append([0] * arity)
    pre.append([p - 1] * arity)
    d = C.limbs_to_ints(C.poseidon_batch(f, arity, C.ints_to_limbs([x for r in pre for x in r])))
    assert d == [R.poseidon_hash(f, r) for r in pre]


def test_tree8_matches_pyref():
    leaves = [R.uniform_fe(2, i, R.PALLAS_Q) for i in range(64)]
    root, lv = C.poseidon_tree8(1, C.ints_to_limbs(leaves), True)
    pl = R.dense_tree_levels(1, leaves)
    assert C.limbs_to_ints(root)[0] == pl[-1][0]
    assert C.limbs_to_ints(lv) == pl[1] + pl[2]


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_curve_and_msm(cn, c):
    order = R.CURVES[cn]["order"]
    assert R.ec_on_curve(cn, R.CURVES[cn]["gen"])
    assert R.ec_mul(cn, order, R.CURVES[cn]["gen"]) is None
    pts = R.synth_bases(cn, 10)
    B = C.synth_bases(c, 10)
    assert C.affine_to_ints(c, B) == pts
    assert C.on_curve(c, B)
    s = [R.uniform_fe(1, i, order) for i in range(10)]
    s[3] = 0
    s[4] = 1
    s[5] = order - 1
    want = R.msm_naive(cn, s, pts)
    S = C.ints_to_limbs(s)
    assert C.jac_to_affine(c, C.msm_naive(c, B, S)) == want
    assert C.jac_to_affine(c, C.msm_pippenger(c, B, S)) == want
    assert C.jac_to_affine(c, C.msm_pippenger(c, B, S, 1)) == want


def test_msm_edge_cases():
    # identity bases (0,0), repeated bases, P + (-P)
    c, cn = 0, "pallas"
    B = C.synth_bases(c, 6)
    B[1] = 0  # identity
    B[3] = B[2]  # repeated point -> doubling inside a bucket
    pts = C.affine_to_ints(c, B)
    ppts = [None if pt == (0, 0) else pt for pt in pts]
    q = R.PALLAS_Q
    s = [5, 7, 9, 9, q - 1, 1]
    B[5] = B[4]  # (q-1)*P + 1*P = identity contribution
    ppts[5] = ppts[4]
    want = R.msm_naive(cn, s, ppts)
    S = C.ints_to_limbs(s)
    assert C.jac_to_affine(c, C.msm_naive(c, B, S)) == want
    assert C.jac_to_affine(c, C.msm_pippenger(c, B, S, 2)) == want
    # all-zero scalars -> identity, encoded (0,0)
    assert C.jac_to_affine(c, C.msm_pippenger(c, B, C.ints_to_limbs([0] * 6))) == (0, 0)


def test_pippenger_vs_naive_vs_dlog_checksum():
    n = 3000
    B = C.synth_bases(0, n)
    S = C.synth_scalars(1, 1, 1, n)  # witness-like distribution
    a = C.jac_to_affine(0, C.msm_naive(0, B, S))
    b = C.jac_to_affine(0, C.msm_pippenger(0, B, S))
    k = C.synth_base_scalars(0, n)
    chk = C.jac_to_affine(0, C.gen_mul(0, C.dot(1, k, S)))
    assert a == b == chk


@pytest.mark.parametrize("f", [0, 1])
def test_ntt(f):
    p = R.modulus(f)
    a = [R.uniform_fe(3, i, p) for i in range(64)]
    fw = C.limbs_to_ints(C.ntt(f, C.ints_to_limbs(a)))
    assert fw == R.dft_naive(p, a) == R.ntt_recursive(p, a)
    assert C.limbs_to_ints(C.ntt(f, C.ints_to_limbs(fw), True)) == a
    w = R.root_of_unity(p, 32)
    assert pow(w, 1 << 32, p) == 1 and pow(w, 1 << 31, p) == p - 1


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_cpu_baseline_msm_matches_the_slow_oracle(cn, c):
    """oracle/msm_fast.c (bench.py's cpu_baseline: pasta-msm-shaped Pippenger) == oracle.c's Pippenger == naive, incl. the edge
    cases of the signed recoding and the exceptional cases of the mixed addition; any thread count."""
    sf = 1 if c == 0 else 0
    q = R.CURVES[cn]["order"]
    for n, dist in ((1, 0), (2, 1), (33, 0), (3000, 1), (1 << 14, 0)):
        B, S = C.synth_bases(c, n), C.synth_scalars(sf, 1, dist, n)
        want = C.jac_to_affine(c, C.msm_pippenger(c, B, S))
        for th in (1, 3, 8):
            assert C.jac_to_affine(c, C.msm_fast(c, B, S, nthreads=th)) == want, (n, th)
    n = 300
    B = C.synth_bases(c, n)
    B[3] = B[2]
    B[5] = 0
    s = [R.uniform_fe(9, i, q) for i in range(n)]
    s[2] = s[3] = 777          # same point, same bucket: the doubling branch
    s[10], s[11] = 555, q - 555
    B[11] = B[10]              # P and -P meet: identity mid-chain
    s[6], s[7], s[8], s[9] = 0, q - 1, 1, (1 << 254) | 0xFFFF
    S = C.ints_to_limbs(s)
    assert C.jac_to_affine(c, C.msm_fast(c, B, S)) == C.jac_to_affine(c, C.msm_naive(c, B, S))
    assert C.jac_to_affine(c, C.msm_fast(c, B, C.ints_to_limbs([0] * n))) == (0, 0)
    assert C.jac_to_affine(c, C.msm_fast(c, B[:0], S[:0])) == (0, 0)

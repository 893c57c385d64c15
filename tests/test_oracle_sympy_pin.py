"""An EXTERNAL pin for the curve arithmetic the MSM oracles rest on: sympy's own elliptic-curve implementation
(sympy.ntheory.elliptic_curve, third-party code, nothing of this repository) against oracle/pyref.py and oracle/oracle.c on
Pallas and Vesta.  The reference holds no Pedersen / MSM vector, so this is the only independent
arithmetic available here: group order, addition, doubling, inverses, scalar multiples and whole multi-scalar sums.  CPU only."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

sympy_ec = pytest.importorskip("sympy.ntheory.elliptic_curve")

CURVES = [("pallas", 0), ("vesta", 1)]


def _curve(cn):
    # both Pasta curves are y^2 = x^3 + 5; Pallas over Fp with q points, Vesta over Fq with p points
    base = R.PALLAS_P if cn == "pallas" else R.PALLAS_Q
    return sympy_ec.EllipticCurve(0, 5, modulus=base), base


def _is_zero(P):
    return int(P.z) == 0  # sympy keeps projective (x : y : z); the point at infinity is (0 : 1 : 0)


def _xy(P):
    return None if _is_zero(P) else (int(P.x), int(P.y))


@pytest.mark.parametrize("cn,c", CURVES)
def test_group_order_and_generator(cn, c):
    E, base = _curve(cn)
    G = E(base - 1, 2)  # (-1, 2): the generator pasta_curves uses on both curves
    order = R.CURVES[cn]["order"]
    assert _is_zero(order * G) and _xy((order - 1) * G) == (base - 1, base - 2)
    assert R.CURVES[cn]["gen"] == (base - 1, 2)


@pytest.mark.parametrize("cn,c", CURVES)
def test_group_law_matches_sympy(cn, c):
    E, base = _curve(cn)
    G = E(base - 1, 2)
    order = R.CURVES[cn]["order"]
    g = R.CURVES[cn]["gen"]
    ks = [1, 2, 3, 0xFFFF, order - 1, order - 2, R.uniform_fe(201, c, order), R.uniform_fe(202, c, order) >> 128]
    pts = [R.ec_mul(cn, k, g) for k in ks]
    for k, pt in zip(ks, pts):
        assert pt == _xy(k * G), k                                  # scalar multiples
        assert C.jac_to_affine(c, C.gen_mul(c, k)) == _xy(k * G)    # the C oracle's
    for (ka, a), (kb, b) in zip(list(zip(ks, pts)), list(zip(ks, pts))[1:] + [(ks[0], pts[0])]):
        assert R.ec_add(cn, a, b) == _xy(ka * G + kb * G)          # addition
    assert R.ec_add(cn, pts[2], pts[2]) == _xy(2 * (3 * G))         # doubling
    assert R.ec_add(cn, pts[2], R.ec_neg(cn, pts[2])) is None and _is_zero(3 * G + (-(3 * G)))
    assert R.ec_add(cn, None, pts[1]) == pts[1]


@pytest.mark.parametrize("cn,c", CURVES)
def test_multi_scalar_sums_match_sympy(cn, c):
    """sum_i s_i P_i three ways - the Python restatement, the C Pippenger and the CPU-baseline Pippenger (msm_fast.c) - against
    sympy's sum of its own scalar multiples; bases are the synthetic key the GPU tests use."""
    E, base = _curve(cn)
    sf = 1 if c == 0 else 0
    n = 20
    B = C.synth_bases(c, n)
    B[5] = 0  # an identity base
    S = C.synth_scalars(sf, 210, 0, n)
    S[3] = 0
    S[4] = C.ints_to_limbs([1])[0]
    scal = C.limbs_to_ints(S)
    pts = [None if pt == (0, 0) else pt for pt in C.affine_to_ints(c, B)]
    acc = None
    for k, pt in zip(scal, pts):
        if pt is not None:
            term = k * E(*pt)  # E(x, y) refuses points off the curve: the synthetic bases are on it
            acc = term if acc is None else acc + term
    want = _xy(acc)
    assert R.msm_naive(cn, scal, pts) == want
    assert C.jac_to_affine(c, C.msm_pippenger(c, B, S)) == want
    assert C.jac_to_affine(c, C.msm_fast(c, B, S)) == want

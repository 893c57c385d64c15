"""CPU: the key-generation oracle's constants checked mathematically (no key bytes exist upstream to compare with)."""
import random

import pytest

from oracle import keygen_ref as K
from oracle import pyref as R


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_isogeny_constants_define_a_homomorphism_onto_the_pasta_curve(curve):
    c = K.CURVE[curve]
    p, a, b = c["p"], c["a"], c["b"]
    rng = random.Random(7)
    pts = []
    while len(pts) < 12:
        x = rng.randrange(p)
        y = K.sqrt_mod((x * x * x + a * x + b) % p, p)
        if y is not None:
            pts.append((x, y))
    imgs = [K.iso_map(curve, pt) for pt in pts]
    for img in imgs:
        assert R.ec_on_curve(curve, img)  # y^2 = x^3 + 5
    for i in range(0, 12, 2):  # phi(P + Q) = phi(P) + phi(Q)
        assert K.iso_map(curve, K.iso_add(curve, pts[i], pts[i + 1])) == R.ec_add(curve, imgs[i], imgs[i + 1])
    assert K.iso_map(curve, K.iso_add(curve, pts[0], pts[0])) == R.ec_add(curve, imgs[0], imgs[0])


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_swu_parameters(curve):
    c = K.CURVE[curve]
    p, a, b, z = c["p"], c["a"], c["b"], c["z"]
    assert not K.is_square(z, p)                                       # RFC 9380 6.6.2: Z is non-square
    x = b * pow(z * a, p - 2, p) % p
    assert K.is_square((x * x * x + a * x + b) % p, p)                 # g(B / (Z A)) is square
    for u in (0, 1, 2, p - 1, 0x1234567890ABCDEF):
        x, y = K.map_to_curve_simple_swu(curve, u)
        assert (y * y - x * x * x - a * x - b) % p == 0 and y % 2 == u % 2


def test_from_label_points_are_on_the_curve_and_deterministic():
    for curve in ("pallas", "vesta"):
        pts = K.from_label(curve, b"ck", 6)
        assert all(R.ec_on_curve(curve, pt) and pt is not None for pt in pts)
        assert K.from_label(curve, b"ck", 3) == pts[:3]                 # a shorter key is a prefix (the XOF is a stream)
        assert K.from_label(curve, b"cl", 2)[0] != pts[0]
    assert len(set(K.from_label("pallas", b"ck", 6))) == 6


def test_library_shake256_is_shake256(hip):
    """The product's host-side XOF (liblurk_hip.so, no GPU needed) against hashlib, across the rate boundary (136 bytes)."""
    import hashlib

    from lurk_beta_amd.msm import shake256

    for msg in (b"", b"ck", b"a" * 135, b"b" * 136, b"c" * 137, bytes(range(256)) * 3):
        for n in (1, 32, 136, 137, 1000):
            assert shake256(msg, n) == hashlib.shake_256(msg).digest(n), (len(msg), n)

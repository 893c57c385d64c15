"""CPU oracle for commitment-key generation: arecibo's `CommitmentKey::setup(b"ck", n)` =
`DlogGroup::from_label(label, n)` as `PublicParams::setup` reaches it from /root/reference/src/proof/nova.rs:196-216.

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py's header).

Un-vendored dependencies restated from their published algorithms:
  * arecibo provider (`from_label`): SHAKE256(label) squeezed 32 bytes per point; point i = hash_to_curve("from_uniform_bytes")
    of its 32 bytes; batch-normalised to affine.
  * pasta_curves 0.5.0 `hashtocurve`: hash_to_field = RFC 9380 expand_message_xmd with BLAKE2b-512 (DST =
    "<domain>-<curve>_XMD:BLAKE2b_SSWU_RO_", 128 output bytes -> two field elements, big-endian wide reduction);
    map_to_curve_simple_swu on the 3-isogenous curve y^2 = x^3 + A x + 1265 with Z = -13 (y's sign = sgn0(u)); the two images
    are added on the isogenous curve and pushed through the degree-3 isogeny to y^2 = x^3 + 5.
PINS: SHAKE256 and BLAKE2b are Python's hashlib here (the product's own implementations are compared with them byte for byte);
the isogenous curves' A and the 13 isogeny coefficients per curve are pasta_curves' published constants and are checked
MATHEMATICALLY (tests/test_oracle_keygen.py): the map sends every point of the isogenous curve onto y^2 = x^3 + 5 and is a group
homomorphism, Z is a non-square and g(B / (Z A)) is a square as RFC 9380 requires.  No key bytes exist upstream to compare with:
"parity unpinned against upstream bytes"."""
from __future__ import annotations

import hashlib

from . import pyref as R

P, Q = R.PALLAS_P, R.PALLAS_Q


def _raw(l):
    return l[0] | l[1] << 64 | l[2] << 128 | l[3] << 192


CURVE = {
    "pallas": dict(p=P, id=b"pallas", a=0x18354A2EB0EA8C9C49BE2D7258370742B74134581A27A59F92BB4B0B657A014B, b=1265, z=P - 13, iso=[_raw(x) for x in [
        [0x775F6034AAAAAAAB, 0x4081775473D8375B, 0xE38E38E38E38E38E, 0x0E38E38E38E38E38],
        [0x8CF863B02814FB76, 0x0F93B82EE4B99495, 0x267C7FFA51CF412A, 0x3509AFD51872D88E],
        [0x0EB64FAEF37EA4F7, 0x380AF066CFEB6D69, 0x98C7D7AC3D98FD13, 0x17329B9EC5253753],
        [0xEEBEC06955555580, 0x8102EEA8E7B06EB6, 0xC71C71C71C71C71C, 0x1C71C71C71C71C71],
        [0xC47F2AB668BCD71F, 0x9C434AC1C96B6980, 0x5A607FCCE0494A79, 0x1D572E7DDC099CFF],
        [0x2AA3AF1EAE5B6604, 0xB4ABF9FB9A1FC81C, 0x1D13BF2A7F22B105, 0x325669BECAECD5D1],
        [0x5AD985B5E38E38E4, 0x7642B01AD461BAD2, 0x4BDA12F684BDA12F, 0x1A12F684BDA12F68],
        [0xC67C31D8140A7DBB, 0x07C9DC17725CCA4A, 0x133E3FFD28E7A095, 0x1A84D7EA8C396C47],
        [0x02E2BE87D225B234, 0x1765E924F7459378, 0x303216CCE1DB9FF1, 0x3FB98FF0D2DDCADD],
        [0x93E53AB371C71C4F, 0x0AC03E8E134EB3E4, 0x7B425ED097B425ED, 0x025ED097B425ED09],
        [0x5A28279B1D1B42AE, 0x5941A3A4A97AA1B3, 0x0790BFB3506DEFB6, 0x0C02C5BCCA0E6B7F],
        [0x4D90AB820B12320A, 0xD976BBFABBC5661D, 0x573B3D7F7D681310, 0x17033D3C60C68173],
        [0x992D30ECFFFFFDE5, 0x224698FC094CF91B, 0x0000000000000000, 0x4000000000000000]]]),
    "vesta": dict(p=Q, id=b"vesta", a=0x267F9B2EE592271A81639C4D96F787739673928C7D01B212C515AD7242EAA6B1, b=1265, z=Q - 13, iso=[_raw(x) for x in [
        [0x43CD42C800000001, 0x0205DD51CFA0961A, 0x8E38E38E38E38E39, 0x38E38E38E38E38E3],
        [0x8B95C6AAF703BCC5, 0x216B8861EC72BD5D, 0xACECF10F5F7C09A2, 0x1D935247B4473D17],
        [0xAEAC67BBEB586A3D, 0xD59D03D23B39CB11, 0xED7EE4A9CDF78F8F, 0x18760C7F7A9AD20D],
        [0xFB539A6F0000002B, 0xE1C521A795AC8356, 0x1C71C71C71C71C71, 0x31C71C71C71C71C7],
        [0xB7284F7EAF21A2E9, 0xA3AD678129B604D3, 0x1454798A5B5C56B2, 0x0A2DE485568125D5],
        [0xF169C187D2533465, 0x30CD6D53DF49D235, 0x0C621DE8B91C242A, 0x14735171EE542778],
        [0x6BEF1642AAAAAAAB, 0x5601F4709A8ADCB3, 0xDA12F684BDA12F68, 0x12F684BDA12F684B],
        [0x8BEE58E5FB81DE63, 0x21D910AEFB03B31D, 0xD6767887AFBE04D1, 0x2EC9A923DA239E8B],
        [0x4986913AB4443034, 0x97A3CA5C24E9EA63, 0x66D1466E9DE10E64, 0x19B0D87E16E25788],
        [0x8F64842C55555533, 0x8BC32D36FB21A6A3, 0x425ED097B425ED09, 0x1ED097B425ED097B],
        [0x58DFECCE86B2745E, 0x06A767BFC35B5BAC, 0x9E7EB64F890A820C, 0x2F44D6C801C1B8BF],
        [0xD43D449776F99D2F, 0x926847FB9DDD76A1, 0x252659BA2B546C7E, 0x3D59F455CAFC7668],
        [0x8C46EB20FFFFFDE5, 0x224698FC0994A8DD, 0x0000000000000000, 0x4000000000000000]]]),
}


def is_square(a: int, p: int) -> bool:
    return a % p == 0 or pow(a, (p - 1) // 2, p) == 1


def sqrt_mod(a: int, p: int):
    """A square root of a (Tonelli-Shanks; p - 1 = 2^32 * odd for both Pasta primes) or None."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    s, q = 0, p - 1
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


# The recalled-from-memory constants of from_label as a parameter block (the oracle's mirror of include/lurk_hip.h:
# lurk_hip_ck_params; the defaults are what the product defaults to).  Tests move one field in BOTH and expect the same new key.
CK_DEFAULTS = dict(xof=0, bytes_per_point=32, domain_prefix="from_uniform_bytes", curve_name_pallas="pallas", curve_name_vesta="vesta",
                   suite="_XMD:BLAKE2b_SSWU_RO_")


def hash_to_field(curve: str, domain_prefix: bytes, msg: bytes, params: dict | None = None) -> tuple[int, int]:
    """pasta_curves hashtocurve::hash_to_field: expand_message_xmd(BLAKE2b-512), two elements."""
    c = CURVE[curve]
    prm = {**CK_DEFAULTS, **(params or {})}
    dst = domain_prefix + b"-" + prm["curve_name_" + curve].encode() + prm["suite"].encode()
    dst_prime = dst + bytes([len(dst)])
    H = lambda data: hashlib.blake2b(data, digest_size=64).digest()
    b0 = H(bytes(128) + msg + bytes([0, 128, 0]) + dst_prime)
    b1 = H(b0 + b"\x01" + dst_prime)
    b2 = H(bytes(x ^ y for x, y in zip(b0, b1)) + b"\x02" + dst_prime)
    return int.from_bytes(b1, "big") % c["p"], int.from_bytes(b2, "big") % c["p"]


def map_to_curve_simple_swu(curve: str, u: int):
    """Affine point on the isogenous curve (never the identity for these parameters)."""
    c = CURVE[curve]
    p, a, b, z = c["p"], c["a"], c["b"], c["z"]
    z_u2 = z * u * u % p
    ta = (z_u2 * z_u2 + z_u2) % p
    num_x1 = b * (ta + 1) % p
    div = a * (z if ta == 0 else (-ta) % p) % p
    inv = pow(div, p - 2, p)
    x1 = num_x1 * inv % p
    gx1 = (x1 * x1 * x1 + a * x1 + b) % p
    if is_square(gx1, p):
        x, y = x1, sqrt_mod(gx1, p)
    else:
        x = z_u2 * x1 % p
        y = sqrt_mod((x * x * x + a * x + b) % p, p)
    if (u % 2) != (y % 2):  # sgn0(u) != sgn0(y)
        y = (-y) % p
    return x, y


def iso_add(curve: str, P1, P2):
    """Addition on the isogenous curve y^2 = x^3 + a x + b (affine, None = identity)."""
    c = CURVE[curve]
    p, a = c["p"], c["a"]
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1 + a) * pow(2 * y1, p - 2, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, p - 2, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def iso_map(curve: str, Pt):
    if Pt is None:
        return None
    c = CURVE[curve]
    p, iso = c["p"], c["iso"]
    x, y = Pt
    nx = (((iso[0] * x + iso[1]) * x + iso[2]) * x + iso[3]) % p
    dx = ((x + iso[4]) * x + iso[5]) % p
    ny = ((((iso[6] * x + iso[7]) * x + iso[8]) * x + iso[9]) * y) % p
    dy = (((x + iso[10]) * x + iso[11]) * x + iso[12]) % p
    if dx == 0 or dy == 0:
        return None  # the kernel of the isogeny
    return nx * pow(dx, p - 2, p) % p, ny * pow(dy, p - 2, p) % p


def hash_to_curve(curve: str, domain_prefix: bytes, msg: bytes, params: dict | None = None):
    u0, u1 = hash_to_field(curve, domain_prefix, msg, params)
    q0, q1 = map_to_curve_simple_swu(curve, u0), map_to_curve_simple_swu(curve, u1)
    return iso_map(curve, iso_add(curve, q0, q1))


def from_label(curve: str, label: bytes, n: int, params: dict | None = None) -> list:
    """arecibo DlogGroup::from_label: n affine points (None = identity, encoded (0,0) at the ABI)."""
    prm = {**CK_DEFAULTS, **(params or {})}
    k = prm["bytes_per_point"]
    stream = (hashlib.shake_128 if prm["xof"] == 1 else hashlib.shake_256)(label).digest(k * n)
    return [hash_to_curve(curve, prm["domain_prefix"].encode(), stream[k * i:k * (i + 1)], prm) for i in range(n)]

"""The folding challenge r (RO row of the scope table): the library's host-side transcript (lurk_beta_amd/csrc/transcript.hip -
arecibo's PoseidonRO over neptune's sponge API, restated [MEM], parity unpinned upstream) against the oracle's independent
restatement (oracle/pyref.py: plain Poseidon schedule at width 25, Python integers).  CPU only: the transcript is host code.
Reference call site: /root/reference/src/proof/nova.rs:282-295 -> arecibo RecursiveSNARK::prove_step -> NIFS::prove."""
import numpy as np

from oracle import coracle as C
from oracle import pyref as R


def test_io_pattern_tag():
    from lurk_beta_amd import _lib

    lib = _lib.load()
    for a, s, d in [(24, 1, 0), (1, 1, 0), (9, 1, 0), (24, 2, 7), (0, 1, 0), (2**31 - 1, 1, 2**32 - 1)]:
        out = np.zeros(2, dtype=np.uint64)
        _lib.check(lib.lurk_hip_nova_ro_pattern_tag(a, s, d, _lib.ptr(out)))
        assert int(out[0]) | int(out[1]) << 64 == R.nova_ro_pattern_tag(a, s, d)
    # by hand: x = 2^128 - 159; value = x (24 + 2^31) + x^2 + 0
    x = (1 << 128) - 159
    assert R.nova_ro_pattern_tag(24, 1) == (x * (24 + (1 << 31)) + x * x) % (1 << 128)


def test_sponge_width_25_constants():
    assert R.round_numbers(R.RO_ARITY) == (8, 59)  # t = 25, standard strength, by neptune's published rule
    assert len(R.round_constants(0, R.RO_ARITY)) == (8 + 59) * 25


def test_ro_squeeze_matches_oracle_on_every_field():
    from lurk_beta_amd import nova_ro_squeeze

    for f in (0, 1, 2):
        p = R.modulus(f)
        for n in (1, 2, 23, 24, 25, 48, 49):  # below, at and across the rate (24): one, two and three permutations
            els = [R.uniform_fe(50 + f, 100 * n + i, p) for i in range(n)]
            for bits in (128, 250):
                assert nova_ro_squeeze(f, els, bits) == R.nova_ro_squeeze(f, els, bits), (f, n, bits)
        assert nova_ro_squeeze(f, [0] * 24, 128) == R.nova_ro_squeeze(f, [0] * 24, 128)
        assert nova_ro_squeeze(f, [p - 1] * 24, 128) == R.nova_ro_squeeze(f, [p - 1] * 24, 128)


def test_ro_rejects_bad_input():
    import pytest

    from lurk_beta_amd import LurkHipError, nova_ro_squeeze

    with pytest.raises(LurkHipError):
        nova_ro_squeeze(0, [R.modulus(0)], 128)  # not canonical
    with pytest.raises(LurkHipError):
        nova_ro_squeeze(0, [1], 0)
    with pytest.raises(LurkHipError):
        nova_ro_squeeze(0, [1], 251)


def _jac(curve, aff):
    """affine (x, y) ints or None -> 96-byte Jacobian in Montgomery form (z = 1, identity z = 0)"""
    f = curve  # base field id of the curve: Pallas -> Fp (0), Vesta -> Fq (1)
    if aff is None:
        return np.zeros(12, dtype=np.uint64)
    return np.concatenate([C.to_mont(f, C.ints_to_limbs([aff[0], aff[1], 1])).reshape(-1)])


def test_nifs_challenge_matches_oracle():
    from lurk_beta_amd import nifs_challenge

    for curve, name in ((0, "pallas"), (1, "vesta")):
        sf = 1 - curve  # scalar field id
        q = R.modulus(sf)
        pts = [R.ec_mul(name, k, R.CURVES[name]["gen"]) for k in (3, 5, 7, 11)]
        for num_io, identity_running in ((2, False), (6, False), (2, True), (0, False)):
            u1 = R.uniform_fe(60, curve, q)
            x1 = [R.uniform_fe(61, i + 10 * curve, q) for i in range(num_io)]
            x2 = [q - 1 - i for i in range(num_io)]  # values above the base modulus on Pallas (q > p): scalar_as_base must reduce
            dig = R.uniform_fe(62, curve, q)
            cw1, ce1 = (None, None) if identity_running else (pts[0], pts[1])
            want = R.nifs_challenge(name, dig, cw1, ce1, u1, x1, pts[2], x2, pts[3])
            got = nifs_challenge(curve, dig, _jac(curve, cw1), _jac(curve, ce1), C.to_mont(sf, C.ints_to_limbs([u1])),
                                 C.to_mont(sf, C.ints_to_limbs(x1)) if num_io else np.zeros((0, 4), dtype=np.uint64), _jac(curve, pts[2]),
                                 C.to_mont(sf, C.ints_to_limbs(x2)) if num_io else np.zeros((0, 4), dtype=np.uint64), _jac(curve, pts[3]))
            assert C.limbs_to_ints(C.from_mont(sf, got.reshape(1, 4)))[0] == want, (name, num_io, identity_running)
            assert want < (1 << 128)
    # NUM_FE_FOR_RO = 24 in Nova for the augmented circuit's two public values: 1 + (3 + 3 + 1 + 2 * 4) + (3 + 2) + 3
    assert len(R.nifs_absorb_list(R.modulus(0), 1, None, None, 1, [1, 2], None, [3, 4], None)) == 24

"""Round 6, verdict item 4: every recalled-from-memory constant of the Nova random oracle and of from_label is a run-time parameter
(include/lurk_hip.h: lurk_hip_ro_params, lurk_hip_ck_params).  CPU only - both are host code of the library.  For every field:
moving it changes r / ck[0], and the product under the moved field equals the oracle's independent restatement under the same move
(oracle/pyref.py: RO_DEFAULTS, oracle/keygen_ref.py: CK_DEFAULTS) - so a Rust host that has to move a field to match arecibo lands
on a value that is still checked.  Reference call sites: /root/reference/src/proof/nova.rs:282-295 (r), :196-216 (ck)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import keygen_ref as K
from oracle import pyref as R

RO_MOVES = [
    dict(arity=16), dict(arity=8), dict(domain_separator=5), dict(absorb_tag_bit=30), dict(num_challenge_bits=96), dict(num_challenge_bits=250),
    dict(item_order=[0, 2, 1, 3]), dict(item_order=[3, 0, 1, 2]), dict(relaxed_order=[2, 0, 1, 3]), dict(relaxed_order=[0, 1, 3, 2]),
    dict(fresh_order=[1, 0]), dict(point_elements=2), dict(relaxed_x_limbs=0), dict(relaxed_x_limbs=3, limb_bits=96), dict(fresh_x_limbs=4),
    dict(limb_bits=32), dict(pattern_absorbs=24), dict(pattern_absorbs=9), dict(squeeze_element=1),
]


def _jac(curve, aff):
    if aff is None:
        return np.zeros(12, dtype=np.uint64)
    return np.concatenate([C.to_mont(curve, C.ints_to_limbs([aff[0], aff[1], 1])).reshape(-1)])


def _case(curve, num_io=6, identity=False):
    name, sf = ("pallas", "vesta")[curve], 1 - curve
    q = R.modulus(sf)
    pts = [R.ec_mul(name, k, R.CURVES[name]["gen"]) for k in (3, 5, 7, 11)]
    u1 = R.uniform_fe(70, curve, q)
    x1 = [R.uniform_fe(71, i + 10 * curve, q) for i in range(num_io)]
    x2 = [q - 1 - i for i in range(num_io)]
    dig = R.uniform_fe(72, curve, q)
    cw1, ce1 = (None, None) if identity else (pts[0], pts[1])
    oracle_args = (name, dig, cw1, ce1, u1, x1, pts[2], x2, pts[3])
    mont = lambda v: C.to_mont(sf, C.ints_to_limbs(v)) if v else np.zeros((0, 4), dtype=np.uint64)
    lib_args = (curve, dig, _jac(curve, cw1), _jac(curve, ce1), mont([u1]), mont(x1), _jac(curve, pts[2]), mont(x2), _jac(curve, pts[3]))
    return oracle_args, lib_args, sf


def test_ro_defaults_are_what_rounds_1_to_5_compiled_in():
    from lurk_beta_amd import params as P

    P.ro_params_set()
    assert P.ro_params_get() == R.RO_DEFAULTS


def test_every_ro_field_moves_r_and_matches_the_oracle_under_the_same_move():
    from lurk_beta_amd import nifs_challenge
    from lurk_beta_amd import params as P

    P.ro_params_set()
    for curve in (0, 1):
        oa, la, sf = _case(curve)
        r0 = C.limbs_to_ints(C.from_mont(sf, nifs_challenge(*la).reshape(1, 4)))[0]
        assert r0 == R.nifs_challenge(*oa)
        l0 = P.nifs_absorb_list(*la)
        assert l0 == R.nifs_absorb_list(R.modulus(curve), *oa[1:])
        seen = {r0}
        for mv in RO_MOVES:
            with P.ro_params(**mv):
                got = C.limbs_to_ints(C.from_mont(sf, nifs_challenge(*la).reshape(1, 4)))[0]
                lst = P.nifs_absorb_list(*la)
            assert got == R.nifs_challenge(*oa, params=mv), (curve, mv)
            assert lst == R.nifs_absorb_list(R.modulus(curve), *oa[1:], params=mv), (curve, mv)
            assert got != r0, (curve, mv)
            seen.add(got)
        assert len(seen) == len(RO_MOVES) + 1  # every move lands somewhere else
        assert P.ro_params_get() == R.RO_DEFAULTS  # the context manager put the defaults back
        assert C.limbs_to_ints(C.from_mont(sf, nifs_challenge(*la).reshape(1, 4)))[0] == r0


def test_ro_moves_with_an_identity_running_instance_and_other_io_counts():
    from lurk_beta_amd import nifs_challenge
    from lurk_beta_amd import params as P

    for curve, num_io, identity in ((0, 2, True), (1, 0, False), (0, 1, False)):
        oa, la, sf = _case(curve, num_io, identity)
        for mv in (dict(point_elements=2), dict(item_order=[2, 3, 1, 0], relaxed_order=[3, 2, 1, 0]), dict(arity=4, fresh_x_limbs=2, limb_bits=128)):
            with P.ro_params(**mv):
                got = C.limbs_to_ints(C.from_mont(sf, nifs_challenge(*la).reshape(1, 4)))[0]
            assert got == R.nifs_challenge(*oa, params=mv), (curve, num_io, identity, mv)


def test_ro_squeeze_follows_the_sponge_fields():
    from lurk_beta_amd import nova_ro_squeeze
    from lurk_beta_amd import params as P

    els = [R.uniform_fe(73, i, R.modulus(0)) for i in range(30)]
    base = nova_ro_squeeze(0, els, 128)
    for mv in (dict(arity=12), dict(domain_separator=1), dict(absorb_tag_bit=29), dict(pattern_absorbs=7), dict(squeeze_element=3)):
        with P.ro_params(**mv):
            got = nova_ro_squeeze(0, els, 128)
        assert got == R.nova_ro_squeeze(0, els, 128, params=mv) and got != base, mv


def test_ro_params_are_validated():
    from lurk_beta_amd import LurkHipError
    from lurk_beta_amd import params as P

    for bad in (dict(arity=1), dict(arity=64), dict(absorb_tag_bit=32), dict(num_challenge_bits=0), dict(num_challenge_bits=251), dict(item_order=[0, 1, 1, 3]),
                dict(relaxed_order=[0, 1, 2, 4]), dict(fresh_order=[1, 1]), dict(point_elements=4), dict(relaxed_x_limbs=17), dict(limb_bits=0),
                dict(squeeze_element=24), dict(pattern_absorbs=1 << 31)):
        with pytest.raises(LurkHipError):
            P.ro_params_set(**bad)
        assert P.ro_params_get() == R.RO_DEFAULTS  # a refused block changes nothing
    # a block of another layout (struct_size) is refused
    import ctypes

    from lurk_beta_amd import _lib

    s = _lib.RoParamsStruct()
    _lib.check(_lib.load().lurk_hip_ro_params_get(ctypes.byref(s)))
    s.struct_size -= 4
    assert _lib.load().lurk_hip_ro_params_set(ctypes.byref(s)) != 0


def _affine_ints(curve, pts):
    out = []
    for row in C.from_mont(curve, np.ascontiguousarray(pts).reshape(-1, 4)).reshape(-1, 2, 4):
        x, y = C.limbs_to_ints(row)
        out.append(None if (x, y) == (0, 0) else (x, y))
    return out


CK_MOVES = [dict(xof=1), dict(bytes_per_point=16), dict(bytes_per_point=64), dict(domain_prefix="from_uniform_byte"), dict(curve_name_pallas="Pallas", curve_name_vesta="Vesta"),
            dict(suite="_XMD:BLAKE2b_SSWU_NU_")]


def test_ck_defaults_and_host_map_match_the_oracle():
    from lurk_beta_amd import params as P

    P.ck_params_set()
    assert P.ck_params_get() == K.CK_DEFAULTS
    for curve, name in ((0, "pallas"), (1, "vesta")):
        for label in (b"ck", b"", b"a longer label than one word"):
            assert _affine_ints(curve, P.ck_from_label_host(curve, label, 6)) == K.from_label(name, label, 6)
    assert P.ck_from_label_host(0, b"ck", 0).shape == (0, 8)


def test_every_ck_field_moves_the_key_and_matches_the_oracle_under_the_same_move():
    from lurk_beta_amd import params as P

    P.ck_params_set()
    for curve, name in ((0, "pallas"), (1, "vesta")):
        k0 = _affine_ints(curve, P.ck_from_label_host(curve, b"ck", 3))
        seen = {k0[0]}
        for mv in CK_MOVES:
            with P.ck_params(**mv):
                got = _affine_ints(curve, P.ck_from_label_host(curve, b"ck", 3))
            assert got == K.from_label(name, b"ck", 3, params=mv), (name, mv)
            assert got[0] != k0[0], (name, mv)
            seen.add(got[0])
        assert len(seen) == len(CK_MOVES) + 1
        assert P.ck_params_get() == K.CK_DEFAULTS


def test_ck_params_are_validated():
    from lurk_beta_amd import LurkHipError
    from lurk_beta_amd import params as P

    for bad in (dict(xof=2), dict(bytes_per_point=0), dict(bytes_per_point=65), dict(bytes_per_point=64, domain_prefix="x" * 31, suite="y" * 31)):
        with pytest.raises(LurkHipError):
            P.ck_params_set(**bad)
        assert P.ck_params_get() == K.CK_DEFAULTS
    with pytest.raises(ValueError):
        P.ck_params_set(domain_prefix="x" * 32)
    with pytest.raises(LurkHipError):
        P.ck_from_label_host(0, b"ck", (1 << 16) + 1)
    with pytest.raises(LurkHipError):
        P.ck_from_label_host(2, b"ck", 1)

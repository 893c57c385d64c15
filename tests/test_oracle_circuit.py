"""CPU: the slot-witness oracle (oracle/circuit_ref.py) against what the reference pins, the product's host-side size
function and trace permutation (compiled with g++ from the very headers the kernels use) against the oracle."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import circuit_ref as CR
from oracle import coracle as C
from oracle import pyref as R
from tests import host_harness as H

vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
TYPES = (("commitment", 3), ("hash4", 4), ("hash6", 6), ("hash8", 8))


def test_bit_decomp_sizes_are_the_references_constants():
    """/root/reference/src/lem/multiframe.rs:495-498: 298 (Pallas = pallas::Scalar), 301 (Vesta), 354 (BN256), 364 (Grumpkin)."""
    assert CR.slot_witness_size(1, "bit_decomp") == 298
    assert CR.slot_witness_size(0, "bit_decomp") == 301
    assert CR.slot_witness_size(2, "bit_decomp") == 354
    assert CR.slot_witness_size(2, "bit_decomp", CR.GRUMPKIN_FR) == 364


def test_library_slot_sizes_match_the_oracle_and_the_reference(hip):
    from lurk_beta_amd import witness as W

    for f, bd in ((1, 298), (0, 301), (2, 354)):
        assert W.slot_witness_size(f, W.SLOT_BIT_DECOMP) == bd
        for name, ar in TYPES:
            assert W.slot_witness_size(f, ar) == CR.slot_witness_size(f, name) == ar + 3 * ((ar + 1) * 8 + R.round_numbers(ar)[1]) + 1
    # the slot part of a BN254 frame: 14 hash4 + 6 hash8 + 1 commitment + 3 bit decompositions (eval.rs:1960-1964) of the 9 119 aux (:1966)
    assert 14 * W.slot_witness_size(2, 4) + 6 * W.slot_witness_size(2, 8) + W.slot_witness_size(2, 3) + 3 * W.slot_witness_size(2, 1) == 7808
    from lurk_beta_amd import LurkHipError

    with pytest.raises(LurkHipError):
        W.slot_witness_size(1, 5)


@pytest.mark.parametrize("f", [0, 1, 2])
def test_oracle_digest_is_the_pinned_hash_and_constraints_hold(f):
    p = R.modulus(f)
    for name, ar in TYPES:
        for pre in ([R.uniform_fe(50 + ar, i, p) for i in range(ar)], [0] * ar, [p - 1] * ar):
            aux, cs = CR.slot_witness(f, name, pre)
            assert aux[:ar] == [x % p for x in pre] and aux[-1] == R.poseidon_hash(f, pre)
            assert not cs.unsatisfied()
            assert len(cs.constraints) == len(aux) - ar  # one constraint per allocated variable after the preimage
    for v in (0, 1, p - 1, p - 2, R.uniform_fe(3, 3, p), (1 << 200) + 12345, p - 3):
        aux, cs = CR.slot_witness(f, "bit_decomp", [v])
        assert not cs.unsatisfied()
        assert all(x in (0, 1) for x in aux[1:])


def test_oracle_reproduces_reference_kats_through_the_circuit():
    with open(os.path.join(os.path.dirname(__file__), "golden", "bn254_poseidon_kats.json")) as fh:
        kats = json.load(fh)
    want = int(kats["hash8_zeros"]["value"], 16) if isinstance(kats["hash8_zeros"], dict) else int(kats["hash8_zeros"], 16)
    assert CR.slot_witness(2, "hash8", [0] * 8)[0][-1] == want  # /root/reference/src/coprocessor/trie/mod.rs:932
    assert CR.slot_witness(2, "commitment", [0, 4, 123])[0][-1] == 0x0DF269CC1A453B80D4694FE3E54F0FF2D68BFA6A6DD6320446AF03691112E89D  # eval_tests.rs:1940-1947


@pytest.mark.parametrize("f", [0, 1, 2])
def test_trace_permutation_of_the_kernels_matches_the_oracle(f):
    """poseidon29_permute_trace + neptune_post_keys (the device code path, compiled for the CPU) == the restated circuit."""
    p = R.modulus(f)
    L = H.lib()
    for name, ar in TYPES:
        pres = [[R.uniform_fe(60 + ar, i * ar + j, p) for j in range(ar)] for i in range(2)] + [[0] * ar, [p - 1] * ar]
        PRE = C.ints_to_limbs([x for r in pres for x in r])
        size = CR.slot_witness_size(f, name)
        out = np.zeros((len(pres) * size, 4), dtype=np.uint64)
        L.hh_poseidon_trace(f, ar, vp(PRE), ctypes.c_size_t(len(pres)), vp(out))
        got = C.limbs_to_ints(out)
        for i, pre in enumerate(pres):
            assert got[i * size:(i + 1) * size] == CR.slot_witness(f, name, pre)[0], (f, name, i)

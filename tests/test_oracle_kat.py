"""Pins the oracle against every golden vector the reference's tests hold for this path
(SURVEY.md section 8c).  CPU only."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests import kat


def test_round_numbers():
    # neptune standard strength: R_F = 8; R_P = 56 (t=4,5,7), 57 (t=9)
    assert [R.round_numbers(a) for a in (3, 4, 6, 8)] == [(8, 56), (8, 56), (8, 56), (8, 57)]


def test_pyref_reproduces_all_reference_kats():
    got = kat.compute_all(lambda pre: R.poseidon_hash(kat.BN, pre))
    for name, val in got.items():
        assert val == kat.golden_int(name), name


def test_c_oracle_reproduces_all_reference_kats():
    def h(pre):
        return C.limbs_to_ints(C.poseidon_batch(kat.BN, len(pre), C.ints_to_limbs(pre)))[0]

    got = kat.compute_all(h)
    for name, val in got.items():
        assert val == kat.golden_int(name), name


def test_trie_path_kat():
    assert R.trie_path(kat.BN, 500, 3) == kat.GOLDEN["trie_path_500_h3"]


def test_trie_helpers_match_recipes():
    assert R.trie_empty_roots(kat.BN, 85)[84] == kat.golden_int("empty_root_85")
    assert R.trie_insert_root(kat.BN, 85, 123, 456) == kat.golden_int("trie_insert_123_456")
    assert R.commit(kat.BN, 0, R.TAG_NIL, R.hash_symbol_path(kat.BN, ["lurk", "nil"])) == kat.golden_int("commit_nil")


def test_unsupported_arity_panics_like_reference():
    # src/hash.rs:19-29: HashArity::from panics on anything but 3,4,6,8
    with pytest.raises(AssertionError):
        R.poseidon_hash(kat.BN, [0] * 5)


def test_structural_identities():
    # src/lem/store.rs:1368-1412 string/symbol hashing are hash4 chains
    f = kat.BN
    hi = R.poseidon_hash(f, [R.TAG_CHAR, ord("h"), R.TAG_STR, R.poseidon_hash(f, [R.TAG_CHAR, ord("i"), R.TAG_STR, 0])])
    assert R.hash_string(f, "hi") == hi
    foo, bar = R.hash_string(f, "foo"), R.hash_string(f, "bar")
    want = R.poseidon_hash(f, [R.TAG_STR, bar, R.TAG_SYM, R.poseidon_hash(f, [R.TAG_STR, foo, R.TAG_SYM, 0])])
    assert R.hash_symbol_path(f, ["foo", "bar"]) == want

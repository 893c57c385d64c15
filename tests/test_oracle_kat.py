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


def test_store_hydration_oracle_in_c_matches_the_recursion():
    """oracle.c: orc_store_hydrate (level by level, all cores) == the node-by-node recursion of pyref on a small list DAG, and
    reproduces the reference's (commit (lambda (x) x)) golden through the StoreHasher layouts (store.rs:29-78, eval_tests.rs:379)."""
    import sys

    from lurk_beta_amd import store_hasher as SH

    big = SH.list_dag(12)
    rec, vals = SH.encode(big)
    got, levels = C.store_hydrate(1, rec, vals)
    memo = {}

    def ref(i):
        if i in memo:
            return memo[i]
        nd = big[i]
        tag = lambda j: 8 if big[j][0] == "comm" else big[j][1]
        if nd[0] == "atom":
            v = nd[2]
        elif nd[0] == "comm":
            v = R.poseidon_hash(1, [nd[1], tag(nd[2]), ref(nd[2])])
        elif nd[0] == "compact":
            a, b, c = nd[2:]
            v = R.poseidon_hash(1, [ref(a), tag(b), ref(b), ref(c)])
        else:
            v = R.poseidon_hash(1, [x for c in nd[2:] for x in (tag(c), ref(c))])
        memo[i] = v
        return v

    sys.setrecursionlimit(10000)
    assert C.limbs_to_ints(got) == [ref(i) for i in range(len(big))]
    assert levels == 7 + 12 + 2  # a symbol's path cells end at level 7, the spine adds one level per cons, then the binding and its commitment
    # the reference's golden: (commit (lambda (x) x)) as a DAG
    nodes = []
    x = SH._symbol_nodes(nodes, ["lurk", "user", "x"])
    nil = SH._symbol_nodes(nodes, ["lurk", "nil"], tag=R.TAG_NIL)
    nodes.append(("tuple2", R.TAG_CONS, x, nil))
    args = len(nodes) - 1
    nodes.append(("atom", R.TAG_ENV, 0))
    nodes.append(("atom", R.TAG_NIL, 0))
    nodes.append(("tuple4", R.TAG_FUN, args, x, len(nodes) - 2, len(nodes) - 1))
    nodes.append(("comm", 0, len(nodes) - 1))
    d, _ = C.store_hydrate(kat.BN, *SH.encode(nodes))
    assert C.limbs_to_ints(d[-1:])[0] == kat.golden_int("commit_lambda_x_x")


def test_host_poseidon_of_the_library_reproduces_all_reference_kats():
    """lurk_hip_poseidon_hash_host (host code of the product library: the store hydration's narrow levels, single hashes) needs no
    device: every reference golden through it, and random batches of every arity and field against the oracle."""
    from lurk_beta_amd import _lib

    lib = _lib.load()

    def h(f, pre):
        a = C.ints_to_limbs(pre)
        out = np.zeros(4, dtype=np.uint64)
        _lib.check(lib.lurk_hip_poseidon_hash_host(f, len(pre), _lib.ptr(a), 1, _lib.ptr(out)))
        return C.limbs_to_ints(out.reshape(1, 4))[0]

    got = kat.compute_all(lambda pre: h(kat.BN, pre))
    assert len(got) == 13
    for name, val in got.items():
        assert val == kat.golden_int(name), name
    for f in (0, 1, 2):
        for arity in (3, 4, 6, 8):
            pre = C.synth_scalars(f, 40 + arity, 0, 37 * arity).reshape(37, arity, 4)
            pre[0] = 0
            pre[1] = C.ints_to_limbs([R.modulus(f) - 1] * arity)
            out = np.zeros((37, 4), dtype=np.uint64)
            _lib.check(lib.lurk_hip_poseidon_hash_host(f, arity, _lib.ptr(pre), 37, _lib.ptr(out)))
            assert np.array_equal(out, C.poseidon_batch(f, arity, pre)), (f, arity)
    bad = C.ints_to_limbs([R.modulus(1)] * 4)  # not canonical
    assert lib.lurk_hip_poseidon_hash_host(1, 4, _lib.ptr(bad), 1, _lib.ptr(np.zeros(4, dtype=np.uint64))) != 0
    assert lib.lurk_hip_poseidon_hash_host(1, 5, _lib.ptr(bad), 1, _lib.ptr(np.zeros(4, dtype=np.uint64))) != 0  # src/hash.rs:19-29

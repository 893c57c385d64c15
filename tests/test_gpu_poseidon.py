"""Parity of the HIP Poseidon path (through the C ABI) against the reference's golden vectors and
the oracle.  Needs an MI355X."""
import ctypes

import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests import kat

pytestmark = pytest.mark.gpu


def test_gpu_reproduces_all_reference_kats(hip):
    """Every BN254 golden vector of the reference, computed by the gfx950 kernel."""
    from lurk_beta_amd import PoseidonCache

    cache = PoseidonCache(kat.BN)
    got = kat.compute_all(lambda pre: cache.compute_hash(pre))
    for name, val in got.items():
        assert val == kat.golden_int(name), name


@pytest.mark.parametrize("f", [0, 1, 2])
@pytest.mark.parametrize("arity", [3, 4, 6, 8])
def test_batch_matches_oracle(hip, f, arity):
    from lurk_beta_amd import poseidon_batch

    p = R.modulus(f)
    n = 1000  # ragged: not a multiple of the 256-thread workgroup
    pre = C.synth_scalars(f, 20 + arity, 0, n * arity).reshape(n, arity, 4)
    pre[0] = 0
    pre[1] = C.ints_to_limbs([p - 1] * arity)
    got = poseidon_batch(f, arity, pre)
    want = C.poseidon_batch(f, arity, pre)
    assert np.array_equal(got, want)


def test_large_batch_grid_stride(hip):
    from lurk_beta_amd import poseidon_batch

    n = 300_000  # more hashes than resident lanes: exercises the grid-stride loop
    pre = C.synth_scalars(1, 31, 1, n * 8).reshape(n, 8, 4)  # witness-like values
    assert np.array_equal(poseidon_batch(1, 8, pre), C.poseidon_batch(1, 8, pre))


def test_empty_batch_and_bad_arity(hip):
    from lurk_beta_amd import LurkHipError, poseidon_batch

    assert poseidon_batch(1, 8, np.zeros((0, 8, 4), dtype=np.uint64)).shape == (0, 4)
    with pytest.raises(LurkHipError):  # hash.rs:19-29: any other arity panics
        poseidon_batch(1, 5, np.zeros((1, 5, 4), dtype=np.uint64))
    with pytest.raises(LurkHipError):
        poseidon_batch(7, 8, np.zeros((1, 8, 4), dtype=np.uint64))


@pytest.mark.parametrize("f,height", [(1, 1), (1, 3), (2, 2), (1, 6)])
def test_tree8_matches_oracle(hip, f, height):
    from lurk_beta_amd import LurkHipError, poseidon_tree8

    n = 8 ** height
    leaves = C.synth_scalars(f, 2, 0, n)
    root, levels = poseidon_tree8(f, leaves, want_levels=True)
    oroot, olevels = C.poseidon_tree8(f, leaves, True)
    assert np.array_equal(root, oroot)
    assert np.array_equal(levels, olevels)
    with pytest.raises(LurkHipError):
        poseidon_tree8(f, np.zeros((12, 4), dtype=np.uint64))  # not a power of 8


def test_tree8_empty_roots_are_the_trie_kats(hip):
    """A dense tree of zero leaves reproduces the trie's empty roots (trie/mod.rs:464-481)."""
    from lurk_beta_amd import poseidon_tree8

    for h, name in ((1, "hash8_zeros"), (2, "empty_root_2"), (3, "empty_root_3"), (4, "empty_root_4")):
        root = poseidon_tree8(kat.BN, np.zeros((8 ** h, 4), dtype=np.uint64))
        assert C.limbs_to_ints(root)[0] == kat.golden_int(name)

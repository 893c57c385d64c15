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


def test_sharded_tree_single_rank_equals_dense_tree(hip):
    """distributed.sharded_tree8_root with no process group (world 1): 8 subtrees + one hash8 == the dense tree."""
    from lurk_beta_amd import poseidon_tree8
    from lurk_beta_amd.distributed import sharded_tree8_root

    for n in (8, 512):
        leaves = C.synth_scalars(1, 2, 0, n)
        assert np.array_equal(sharded_tree8_root(1, leaves), poseidon_tree8(1, leaves))
    with pytest.raises(ValueError):
        sharded_tree8_root(1, C.synth_scalars(1, 2, 0, 16))


def test_tree8_empty_roots_are_the_trie_kats(hip):
    """A dense tree of zero leaves reproduces the trie's empty roots (trie/mod.rs:464-481)."""
    from lurk_beta_amd import poseidon_tree8

    for h, name in ((1, "hash8_zeros"), (2, "empty_root_2"), (3, "empty_root_3"), (4, "empty_root_4")):
        root = poseidon_tree8(kat.BN, np.zeros((8 ** h, 4), dtype=np.uint64))
        assert C.limbs_to_ints(root)[0] == kat.golden_int(name)


def test_trie_mirror_reproduces_reference_kats(hip):
    """coprocessor::trie::Trie through the GPU hasher: empty roots, path, insert, lookup
    (trie/mod.rs:925-1017, eval_tests.rs:3868,3904)."""
    from lurk_beta_amd.trie import Trie

    t3 = Trie(kat.BN, height=3)
    assert [t3.empty_root_for_height(h) for h in (0, 1, 2, 3)] == [0, kat.golden_int("hash8_zeros"), kat.golden_int("empty_root_2"),
                                                                    kat.golden_int("empty_root_3")]
    assert t3.leaves() == 512 and t3.path(500) == kat.GOLDEN["trie_path_500_h3"]
    t = Trie(kat.BN)  # StandardTrie: arity 8, height 85
    assert t.root == kat.golden_int("empty_root_85")
    assert t.lookup(123) is None
    assert t.insert(123, 456) is False
    assert t.root == kat.golden_int("trie_insert_123_456")
    assert t.lookup(123) == 456 and t.lookup(124) is None
    assert t.insert(123, 789) is True and t.lookup(123) == 789
    # the same operations over Pallas Fq agree with the oracle's recursion
    tp = Trie(1, height=5)
    tp.insert(77, 1234)
    assert tp.root == R.trie_insert_root(1, 5, 77, 1234)


def test_store_hydration_level_batches(hip):
    """StoreHasher layouts hashed level by level on the GPU == node-by-node recursion in the oracle
    (store.rs:29-78; KAT: (commit (lambda (x) x)), eval_tests.rs:379)."""
    from lurk_beta_amd import PoseidonCache
    from lurk_beta_amd.store_hasher import hydrate

    def string_nodes(nodes, s):  # Str cells, terminator (Str, 0)
        nodes.append(("atom", R.TAG_STR, 0))
        cur = len(nodes) - 1
        for ch in reversed(s):
            nodes.append(("atom", R.TAG_CHAR, ord(ch)))
            nodes.append(("tuple2", R.TAG_STR, len(nodes) - 1, cur))
            cur = len(nodes) - 1
        return cur

    def symbol_nodes(nodes, path, tag=R.TAG_SYM):
        nodes.append(("atom", R.TAG_SYM, 0))
        cur = len(nodes) - 1
        for k, name in enumerate(path):
            sn = string_nodes(nodes, name)
            nodes.append(("tuple2", tag if k == len(path) - 1 else R.TAG_SYM, sn, cur))
            cur = len(nodes) - 1
        return cur

    nodes = []
    x = symbol_nodes(nodes, ["lurk", "user", "x"])
    nil = symbol_nodes(nodes, ["lurk", "nil"], tag=R.TAG_NIL)
    nodes.append(("tuple2", R.TAG_CONS, x, nil))          # (x)
    args = len(nodes) - 1
    nodes.append(("atom", R.TAG_ENV, 0))                  # empty env
    env = len(nodes) - 1
    nodes.append(("atom", R.TAG_NIL, 0))                  # dummy
    dummy = len(nodes) - 1
    nodes.append(("tuple4", R.TAG_FUN, args, x, env, dummy))
    fun = len(nodes) - 1
    nodes.append(("comm", 0, fun))
    digests = hydrate(PoseidonCache(kat.BN), nodes)
    assert digests[-1] == kat.golden_int("commit_lambda_x_x")
    assert digests[nil] == R.hash_symbol_path(kat.BN, ["lurk", "nil"])
    # compact (env binding) layout
    nodes2 = [("atom", R.TAG_SYM, 11), ("atom", R.TAG_NUM, 22), ("atom", R.TAG_ENV, 0), ("compact", R.TAG_ENV, 0, 1, 2)]
    d2 = hydrate(PoseidonCache(1), nodes2)
    assert d2[3] == R.poseidon_hash(1, [11, R.TAG_NUM, 22, 0])
    # the same DAGs through the C ABI entry point (all levels on the device, no per-level round trip)
    from lurk_beta_amd.store_hasher import hydrate_device

    assert hydrate_device(kat.BN, nodes) == digests
    assert hydrate_device.last_levels == max(len("lurk"), len("user"), len("nil")) + 3 + 3  # chars + symbol path + cons, fun, comm
    assert hydrate_device(1, nodes2) == d2
    # a few-thousand-node DAG: a list of 400 distinct symbols, a tuple3 and compact bindings on top; node-by-node oracle recursion
    import time

    big = []
    syms = [symbol_nodes(big, ["lurk", "user", "sym%d" % k]) for k in range(400)]
    big.append(("atom", R.TAG_NIL, 0))
    lst = len(big) - 1
    for sy in reversed(syms):
        big.append(("tuple2", R.TAG_CONS, sy, lst))
        lst = len(big) - 1
    big.append(("tuple3", R.TAG_THUNK, syms[0], syms[1], lst))
    big.append(("compact", R.TAG_ENV, syms[2], syms[3], lst))
    big.append(("comm", 12345, len(big) - 2))
    t0 = time.perf_counter()
    got = hydrate_device(1, big)
    dt = time.perf_counter() - t0
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

    import sys

    sys.setrecursionlimit(10000)
    for i in (len(big) - 1, len(big) - 2, len(big) - 3, lst, syms[17], 5):
        assert got[i] == ref(i), i
    print("store hydrate: %d nodes, %d levels, %.1f ms" % (len(big), hydrate_device.last_levels, dt * 1e3))


def test_store_hydration_deep_and_wide_dags_every_digest(hip):
    """Both shapes bench.py --workload store_hydrate times, every digest against the oracle's hydration (oracle.c: orc_store_hydrate,
    itself = the node-by-node recursion, tests/test_oracle_kat.py): the DEEP list DAG (a 400-cell spine: narrow levels go to the
    library's host Poseidon, wide ones stay on the device, the digest array changes sides once) and a WIDE one (~5 x 10^5 nodes,
    ~30 levels, all on the device but its top)."""
    from lurk_beta_amd import store_hasher as SH

    for f, dag in ((1, SH.list_dag(400)), (kat.BN, SH.list_dag(37)), (1, SH.wide_dag(12000))):
        rec, vals = SH.encode(dag)
        want, want_levels = C.store_hydrate(f, rec, vals)
        got, levels = SH.hydrate_records(f, rec, vals)
        assert levels == want_levels
        assert np.array_equal(got, want), (f, len(dag))
    # a DAG whose levels alternate between wide and narrow: the digest array changes sides at every level
    nodes = []
    prev = [SH._symbol_nodes(nodes, ["a%d" % j]) for j in range(64)]
    for rnd in range(6):
        nodes.append(("tuple2", 1, prev[0], prev[1]))                      # a narrow level (one node) ...
        top = len(nodes) - 1
        prev = [top] + prev[2:]
        nxt = []
        for j in range(len(prev)):                                         # ... then a wide one that depends on it
            nodes.append(("tuple2", 1, prev[j], top))
            nxt.append(len(nodes) - 1)
        prev = nxt
    rec, vals = SH.encode(nodes)
    want, _ = C.store_hydrate(1, rec, vals)
    got, _ = SH.hydrate_records(1, rec, vals)
    assert np.array_equal(got, want)


def test_full_size_tree_2_24(hip):
    """BASELINE configs[2] size (8^8 = 2^24 leaves, 2 396 745 hash8) through size-independent properties:
    (i) over BN254, a tree of zero leaves must reproduce element 8 of the trie's empty-root chain - the chain the
        reference's golden vectors pin at elements 1..4 and 85 (trie/mod.rs:464-481, eval_tests.rs:3868);
    (ii) over Pallas Fq with random leaves, 600 nodes sampled from every level satisfy node = hash8(children)
        when recomputed by the oracle, and the root equals the hash8 of the 8 subtree roots."""
    import torch

    from lurk_beta_amd import _lib, synth

    lib = _lib.load()
    n = 8 ** 8
    stream = torch.cuda.current_stream().cuda_stream
    d_levels = torch.empty(((n - 1) // 7, 4), dtype=torch.int64, device="cuda")
    # (i)
    d_zero = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    _lib.check(lib.lurk_hip_poseidon_tree8_dev(kat.BN, _lib.ptr(d_zero), n, _lib.ptr(d_levels), _lib.ptr(stream)))
    torch.cuda.synchronize()
    cur, chain = 0, []
    for _ in range(8):
        cur = R.poseidon_hash(kat.BN, [cur] * 8)
        chain.append(cur)
    assert chain[1] == kat.golden_int("empty_root_2") and chain[3] == kat.golden_int("empty_root_4")
    assert C.limbs_to_ints(d_levels[-1:].cpu().numpy().view(np.uint64))[0] == chain[7]
    del d_zero
    # (ii)
    f = 1
    d_leaves = synth.scalars(f, 2, 0, n)
    _lib.check(lib.lurk_hip_poseidon_tree8_dev(f, _lib.ptr(d_leaves), n, _lib.ptr(d_levels), _lib.ptr(stream)))
    torch.cuda.synchronize()
    rng = np.random.default_rng(4)
    starts, sizes, off, m = [], [], 0, n // 8
    while m >= 1:
        starts.append(off)
        sizes.append(m)
        off += m
        m //= 8
    pre, want_idx = [], []
    for lvl, (st, sz) in enumerate(zip(starts, sizes)):
        for j in rng.integers(0, sz, 75).tolist():
            child = d_leaves[8 * j:8 * j + 8] if lvl == 0 else d_levels[starts[lvl - 1] + 8 * j: starts[lvl - 1] + 8 * j + 8]
            pre.append(child.cpu().numpy().view(np.uint64))
            want_idx.append(st + j)
    got = d_levels[torch.tensor(want_idx, device="cuda")].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, C.poseidon_batch(f, 8, np.stack(pre)))
    top8 = d_levels[starts[-2]:starts[-2] + 8].cpu().numpy().view(np.uint64)
    assert np.array_equal(d_levels[-1].cpu().numpy().view(np.uint64), C.poseidon_batch(f, 8, top8.reshape(1, 8, 4))[0])

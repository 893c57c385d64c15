"""CPU-only: the library's host/device arithmetic headers (compiled with g++ by tests/host_harness)
and its host-side Poseidon parameter generation, against the oracle."""
import ctypes

import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from tests import host_harness as H

vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731


def _op(f, o, a, b):
    A, B = C.ints_to_limbs(a), C.ints_to_limbs(b)
    O = np.zeros_like(A)
    H.lib().hh_fe_op(f, o, vp(A), vp(B), vp(O), ctypes.c_size_t(len(a)))
    return C.limbs_to_ints(O)


@pytest.mark.parametrize("f", [0, 1, 2])
def test_field_ops(f):
    p = R.modulus(f)
    Rm = 1 << 256
    Ri = pow(Rm, -1, p)
    a = [R.uniform_fe(9, i, p) for i in range(300)] + [0, 1, p - 1, p - 1, 0]
    b = [R.uniform_fe(10, i, p) for i in range(300)] + [0, p - 1, p - 1, 1, 5]
    assert _op(f, 0, a, b) == [x * y * Ri % p for x, y in zip(a, b)]
    assert _op(f, 8, a, b) == [x * y * Ri % p for x, y in zip(a, b)]  # fe_mul_fips (device schedule)
    assert _op(f, 9, a, b) == [x * y * Ri % p for x, y in zip(a, b)]  # fe_mul_cios
    assert _op(f, 1, a, b) == [(x + y) % p for x, y in zip(a, b)]
    assert _op(f, 2, a, b) == [(x - y) % p for x, y in zip(a, b)]
    assert _op(f, 3, a, b) == [x * x * Ri % p for x in a]
    assert _op(f, 5, a, b) == [x * Rm % p for x in a]
    assert _op(f, 6, a, b) == [x * Ri % p for x in a]
    assert _op(f, 7, a, b) == [(-x) % p for x in a]
    aa = a[:10] + [0]
    assert _op(f, 4, aa, aa) == [(pow(x * Ri % p, -1, p) * Rm % p if x else 0) for x in aa]


@pytest.mark.parametrize("f", [0, 1, 2])
@pytest.mark.parametrize("arity", [3, 4, 6, 8])
def test_poseidon_params_and_sparse_schedule(f, arity):
    L = H.lib()
    rf, rp = ctypes.c_int(), ctypes.c_int()
    rc = np.zeros((700, 4), dtype=np.uint64)
    mds = np.zeros((81, 4), dtype=np.uint64)
    n = L.hh_poseidon_params(f, arity, ctypes.byref(rf), ctypes.byref(rp), vp(rc), vp(mds))
    assert (rf.value, rp.value) == R.round_numbers(arity)
    assert C.limbs_to_ints(rc[:n]) == list(R.round_constants(f, arity))
    assert C.limbs_to_ints(mds[: (arity + 1) ** 2]) == [x for r in R.mds_matrix(f, arity) for x in r]
    p = R.modulus(f)
    pre = [[R.uniform_fe(2, i * arity + j, p) for j in range(arity)] for i in range(4)] + [[0] * arity, [p - 1] * arity]
    PRE = C.ints_to_limbs([x for r in pre for x in r])
    want = C.limbs_to_ints(C.poseidon_batch(f, arity, PRE))
    for mode in (0, 1, 2):  # 0 = sparse schedule, 1 = plain schedule, 2 = sparse schedule on the radix-2^29 layer (what the kernel runs)
        out = np.zeros((len(pre), 4), dtype=np.uint64)
        L.hh_poseidon(f, arity, mode, vp(PRE), ctypes.c_size_t(len(pre)), vp(out))
        assert C.limbs_to_ints(out) == want, (f, arity, mode)


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_xyzz_group_law_with_exceptional_cases(cn, c):
    L = H.lib()
    B = C.synth_bases(c, 12)
    B[3] = B[2]          # same point twice in a row -> doubling branch of madd/add
    B[5] = 0             # identity base
    B[7] = B[6]          # with opposite signs below -> P + (-P) = identity mid-chain
    pts = [None if pt == (0, 0) else pt for pt in C.affine_to_ints(c, B)]
    signs = np.array([0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1, 0], dtype=np.uint32)
    want = None
    for pt, s in zip(pts, signs):
        want = R.ec_add(cn, want, R.ec_neg(cn, pt) if s else pt)
    for mode in (0, 1, 3, 4, 5):  # 3 = radix-2^29 accumulator (the bucket-accumulation kernel's inner loop); 5 = the finalize stage's sum of partials
        out = np.zeros(8, dtype=np.uint64)
        L.hh_curve_sum(c, mode, vp(B), vp(signs), ctypes.c_size_t(12), vp(out))
        assert C.affine_to_ints(c, out)[0] == want, mode
    # chain that ends on the identity
    B2 = np.concatenate([B[:1], B[:1]])
    for mode in (0, 3, 4, 5):
        out = np.zeros(8, dtype=np.uint64)
        L.hh_curve_sum(c, mode, vp(B2), vp(np.array([0, 1], dtype=np.uint32)), ctypes.c_size_t(2), vp(out))
        assert C.affine_to_ints(c, out)[0] == (0, 0)
    # partials that are equal (the doubling branch of the general addition), opposite (the identity mid-sum) and the identity itself
    B6 = np.concatenate([B[:3], B[:3], B[8:11], B[8:11], B[:3], B[5:6], B[5:6], B[5:6]])
    s6 = np.array([0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0], dtype=np.uint32)
    o0, o5 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    L.hh_curve_sum(c, 0, vp(B6), vp(s6), ctypes.c_size_t(18), vp(o0))
    L.hh_curve_sum(c, 5, vp(B6), vp(s6), ctypes.c_size_t(18), vp(o5))
    assert (o0 == o5).all() and o0.any()
    # starts with a negated point, then doubles it, then keeps going
    B3 = np.concatenate([B[:1], B[:1], B[1:4]])
    s3 = np.array([1, 1, 0, 1, 1], dtype=np.uint32)
    outs = []
    for mode in (0, 3, 4):
        out = np.zeros(8, dtype=np.uint64)
        L.hh_curve_sum(c, mode, vp(B3), vp(s3), ctypes.c_size_t(5), vp(out))
        outs.append(C.affine_to_ints(c, out)[0])
    assert outs[0] == outs[1] == outs[2]
    # small-scalar multiples
    ks = np.array([0, 1, 2, 3, 17, 255, 32768, 65535, 1, 0, 5, 6], dtype=np.uint32)
    want = None
    for pt, k in zip(pts, ks):
        want = R.ec_add(cn, want, R.ec_mul(cn, int(k), pt) if pt else None)
    out = np.zeros(8, dtype=np.uint64)
    L.hh_curve_sum(c, 2, vp(B), vp(ks), ctypes.c_size_t(12), vp(out))
    assert C.affine_to_ints(c, out)[0] == want


@pytest.mark.parametrize("f", [0, 1, 2])
@pytest.mark.parametrize("T", [3, 5, 9])
def test_inner_product_with_single_reduction(f, T):
    """fe_dot (dot_mac / dot_finish): sum a_i*b_i / R mod p, including the worst case (all p-1)."""
    p = R.modulus(f)
    Ri = pow(1 << 256, -1, p)
    cases = [([R.uniform_fe(40 + k, i, p) for i in range(T)], [R.uniform_fe(50 + k, i, p) for i in range(T)]) for k in range(20)]
    cases.append(([p - 1] * T, [p - 1] * T))
    cases.append(([0] * T, [p - 1] * T))
    cases.append(([1] + [0] * (T - 1), [1] + [0] * (T - 1)))
    for a, b in cases:
        A, B = C.ints_to_limbs(a), C.ints_to_limbs(b)
        O = np.zeros(4, dtype=np.uint64)
        H.lib().hh_fe_dot(f, T, vp(A), vp(B), vp(O))
        assert C.limbs_to_ints(O)[0] == sum(x * y for x, y in zip(a, b)) * Ri % p


@pytest.mark.parametrize("f", [0, 1, 2])
def test_radix29_layer(f):
    """field29.cuh (9 x 29-bit limbs, R' = 2^261, lazy reduction) against big integers; I/O in the C-ABI form."""
    p = R.modulus(f)
    Rm = 1 << 256
    Ri = pow(Rm, -1, p)
    a = [R.uniform_fe(60, i, p) for i in range(200)] + [0, 1, p - 1, p - 1, 0, 5]
    b = [R.uniform_fe(61, i, p) for i in range(200)] + [0, p - 1, p - 1, 1, 7, p - 2]
    A, B = C.ints_to_limbs(a), C.ints_to_limbs(b)

    def run(op):
        O = np.zeros_like(A)
        H.lib().hh_f29_op(f, op, vp(A), vp(B), vp(O), ctypes.c_size_t(len(a)))
        return C.limbs_to_ints(O)

    assert run(0) == a                                                       # round trip
    assert run(1) == [x * y * Ri % p for x, y in zip(a, b)]                  # Montgomery product semantics preserved
    assert run(2) == [(x + y) % p for x, y in zip(a, b)]
    assert run(3) == [(x - y) % p for x, y in zip(a, b)]
    assert run(4) == [(x - y) * (x - y) * Ri % p for x, y in zip(a, b)]
    assert run(5) == [(x * y * Ri - x + y) % p for x, y in zip(a, b)]


@pytest.mark.parametrize("c", [0, 1])
def test_radix29_accumulator_long_chains(c):
    """xyzz29_madd against xyzz_madd over long random chains (bounds are asserted inside the harness build)."""
    L = H.lib()
    n = 3000
    B = C.synth_bases(c, n)
    rng = np.random.default_rng(5)
    signs = rng.integers(0, 2, n).astype(np.uint32)
    for lo, hi in [(0, 64), (64, 1000), (1000, 3000)]:
        outs = []
        for mode in (0, 3, 4):
            out = np.zeros(8, dtype=np.uint64)
            L.hh_curve_sum(c, mode, vp(B[lo:hi].copy()), vp(signs[lo:hi].copy()), ctypes.c_size_t(hi - lo), vp(out))
            outs.append(C.affine_to_ints(c, out)[0])
        assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("f", [0, 1])
def test_radix29_partial_reduction(f):
    """f29_reduce: same residue, result < 2^255.1 with tight limbs, for extreme limb patterns."""
    p = R.modulus(f)
    M = (1 << 29) - 1
    rng = np.random.default_rng(9)
    rows = [[M] * 8 + [(1 << 31) - 1], [0] * 9, [0] * 8 + [1 << 22], [M] * 8 + [(1 << 22) - 1], [0] * 8 + [(1 << 31) - 1],
            [0] * 8 + [(1 << 22) + 5], [1] + [0] * 7 + [1 << 23]]
    for _ in range(300):
        rows.append([int(x) for x in rng.integers(0, M + 1, 8)] + [int(rng.integers(0, 1 << 31))])
    I = np.array(rows, dtype=np.uint32)
    O = np.zeros_like(I)
    H.lib().hh_f29_reduce(f, vp(I), vp(O), ctypes.c_size_t(len(rows)))
    for r_in, r_out in zip(rows, O.tolist()):
        vin = sum(x << (29 * i) for i, x in enumerate(r_in))
        vout = sum(x << (29 * i) for i, x in enumerate(r_out))
        assert vout % p == vin % p
        assert all(x <= M for x in r_out[:8]) and r_out[8] < (1 << 23) + 1
        assert vout < (1 << 255) + (1 << 254)


def test_bound_assertions_are_compiled_in_and_the_known_hard_window_passes():
    """The harness exists to run the radix-2^29 code with every limb / accumulator bound asserted: make sure the switch reaches
    the headers.  Points 640..703 of the synthetic Pallas key are the chain on which an accumulator without the Y3 reduction went
    wrong (affine-accumulator first addition: Y3 above 64p): the shipped schedule must agree with the plain one on it."""
    L = H.lib()
    L.hh_f29_checks_active.restype = ctypes.c_int
    assert L.hh_f29_checks_active() == 1
    B = C.synth_bases(0, 704)[640:].copy()
    for signs in (np.zeros(64, dtype=np.uint32), np.ones(64, dtype=np.uint32), (np.arange(64) % 2).astype(np.uint32)):
        outs = []
        for mode in (0, 3, 4):
            out = np.zeros(8, dtype=np.uint64)
            L.hh_curve_sum(0, mode, vp(B), vp(signs), ctypes.c_size_t(64), vp(out))
            outs.append(C.affine_to_ints(0, out)[0])
        assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("c", [0, 1])
def test_accumulator_stress_under_bound_assertions(c):
    """40 000 points in task-sized chains (64 additions, the affine-accumulator first step each time, random signs) through the
    kernel's schedule (mode 3) and the general-addition-only one (mode 4), every lazy bound asserted: no abort, same sums."""
    L = H.lib()
    n = 40000
    B = C.synth_bases(c, n)
    rng = np.random.default_rng(77 + c)
    for w in range(0, n - 64, 64):
        Bc = np.ascontiguousarray(B[w:w + 64])
        signs = rng.integers(0, 2, 64).astype(np.uint32)
        o3, o4 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        L.hh_curve_sum(c, 3, vp(Bc), vp(signs), ctypes.c_size_t(64), vp(o3))
        L.hh_curve_sum(c, 4, vp(Bc), vp(signs), ctypes.c_size_t(64), vp(o4))
        assert np.array_equal(o3, o4), w


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_radix29_reduction_tree_with_bound_assertions(cn, c):
    """xyzz29_add (the node of the small-commitment path's reduction trees, msm_small.hip) under the bound assertions: signed bases
    dealt to 1..64 accumulators and summed by the xor butterfly, against the Python group law; with the exceptional meetings
    (equal partial sums -> doubling, opposite ones -> identity, identity operands) forced into the tree."""
    L = H.lib()
    assert L.hh_f29_checks_active() == 1
    rng = np.random.default_rng(5 + c)
    n = 200
    B = C.synth_bases(c, n)
    pts = [None if pt == (0, 0) else pt for pt in C.affine_to_ints(c, B)]
    for lanes in (1, 2, 4, 16, 64):
        signs = rng.integers(0, 2, n).astype(np.uint32)
        want = None
        for pt, s in zip(pts, signs):
            want = R.ec_add(cn, want, R.ec_neg(cn, pt) if s else pt)
        out = np.zeros(8, dtype=np.uint64)
        L.hh_curve_tree(c, vp(B), vp(signs), ctypes.c_size_t(n), lanes, vp(out))
        assert C.affine_to_ints(c, out)[0] == (want or (0, 0)), lanes
    # 4 lanes: lanes 0 and 1 hold the same point (doubling at level 1), lanes 2 and 3 opposite points (identity at level 1)
    B4 = np.concatenate([B[:1], B[:1], B[1:2], B[1:2]])
    s4 = np.array([0, 0, 0, 1], dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint64)
    L.hh_curve_tree(c, vp(B4), vp(s4), ctypes.c_size_t(4), 4, vp(out))
    assert C.affine_to_ints(c, out)[0] == R.ec_add(cn, pts[0], pts[0])
    # all four equal: doubling at both levels; 8 lanes with only 3 points: identity operands in the tree
    B5 = np.concatenate([B[:1]] * 4)
    out = np.zeros(8, dtype=np.uint64)
    L.hh_curve_tree(c, vp(B5), vp(np.zeros(4, dtype=np.uint32)), ctypes.c_size_t(4), 4, vp(out))
    assert C.affine_to_ints(c, out)[0] == R.ec_mul(cn, 4, pts[0])
    out = np.zeros(8, dtype=np.uint64)
    L.hh_curve_tree(c, vp(B[:3]), vp(np.zeros(3, dtype=np.uint32)), ctypes.c_size_t(3), 8, vp(out))
    assert C.affine_to_ints(c, out)[0] == R.ec_add(cn, R.ec_add(cn, pts[0], pts[1]), pts[2])
    # everything cancels
    B6 = np.concatenate([B[:8], B[:8]])
    s6 = np.array([0] * 8 + [1] * 8, dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint64)
    L.hh_curve_tree(c, vp(B6), vp(s6), ctypes.c_size_t(16), 8, vp(out))
    assert C.affine_to_ints(c, out)[0] == (0, 0)
    # a long stress: 20 000 points through 64 lanes (312 mixed additions per lane, then 6 levels), checked against the 32-bit group law
    n7 = 20000
    B7 = C.synth_bases(c, n7)
    s7 = rng.integers(0, 2, n7).astype(np.uint32)
    o29, o32 = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    L.hh_curve_tree(c, vp(B7), vp(s7), ctypes.c_size_t(n7), 64, vp(o29))
    L.hh_curve_sum(c, 0, vp(B7), vp(s7), ctypes.c_size_t(n7), vp(o32))
    assert np.array_equal(o29, o32)


@pytest.mark.parametrize("c", [6, 8, 13, 16, 17, 18, 20])
def test_signed_digits_register_walk_equals_indexed_recoding(c):
    """msm_digit_next (the walk the sort kernels and the small-commitment kernel use: no limb is indexed by a run-time value) against
    msm_digit_step, and both against the definition: sum_w d_w 2^(c w) = k with |d_w| <= 2^(c-1)."""
    W = (256 + c - 1) // c
    q = R.CURVES["pallas"]["order"]
    ks = [0, 1, q - 1, (1 << 254) - 1, 1 << (c - 1), (1 << (c - 1)) + 1, (1 << c) - 1, int("55" * 32, 16) % q, int("aa" * 31, 16)]
    ks += [R.uniform_fe(900 + c, i, q) for i in range(40)]
    for k in ks:
        s = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
        a, b = np.zeros(W, dtype=np.uint32), np.zeros(W, dtype=np.uint32)
        H.lib().hh_msm_digits(vp(s), c, vp(a), vp(b))
        assert (a == b).all(), (c, hex(k))
        if c != 18:  # the compile-time recoding of the sort kernels (msm_digits_ct<C>; the library instantiates 16 and 20)
            ct = np.zeros(W, dtype=np.uint32)
            assert H.lib().hh_msm_digits_ct(vp(s), c, vp(ct)) == W
            assert (ct == a).all(), (c, hex(k))
        total = 0
        for w in range(W):
            mag, neg = int(a[w]) & 0x7FFFFFFF, int(a[w]) >> 31
            assert mag <= 1 << (c - 1)
            total += (-mag if neg else mag) << (c * w)
        assert total == k, (c, hex(k))


@pytest.mark.parametrize("f", [0, 1, 2])
def test_host_inverse_equals_the_exponentiation(f):
    """field.cuh: on the host fe_inv is the binary extended Euclidean algorithm (the device keeps a^(p-2)); both on 0, 1, 2, p - 1, the
    powers of two, values around the limb boundaries, (p +- 1) / 2, and 6 000 uniform elements - and x * x^-1 = 1 in Python integers."""
    L = H.lib()
    p = R.modulus(f)
    vals = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, (1 << 192) + 1, (1 << 253) - 1, (1 << 253) + 12345]
    vals += [1 << k for k in range(1, 254)] + [(p - (1 << k)) % p for k in range(0, 254, 7)]
    vals += [R.uniform_fe(400 + f, i, p) for i in range(6000)]
    vals = [v % p for v in vals]
    a = C.to_mont(f, C.ints_to_limbs(vals))
    oh, op_ = np.zeros_like(a), np.zeros_like(a)
    L.hh_fe_inv_both(f, vp(a), vp(oh), vp(op_), ctypes.c_size_t(len(vals)))
    assert (oh == op_).all()
    inv = C.limbs_to_ints(C.from_mont(f, oh))
    for v, w in zip(vals[:400], inv[:400]):
        assert (v * w) % p == (1 if v else 0)
    # raw 256-bit patterns that are NOT canonical (a caller's garbage in a point's coordinates): the host inverse must return - the
    # inverse of the residue, 0 for a multiple of p - and never loop
    raw = [p, 2 * p, 3 * p, p + 5, 2 * p + 7, (1 << 256) - 1, (1 << 255) + 3]
    raw = [v for v in raw if v < 1 << 256]
    a = C.ints_to_limbs(raw)
    oh, op_ = np.zeros_like(a), np.zeros_like(a)
    L.hh_fe_inv_both(f, vp(a), vp(oh), vp(op_), ctypes.c_size_t(len(raw)))
    R2 = pow(2, 512, p)
    assert C.limbs_to_ints(oh) == [(pow(v % p, -1, p) * R2) % p if v % p else 0 for v in raw]


@pytest.mark.parametrize("f", [0, 1, 2])
@pytest.mark.parametrize("np_", [4, 2])
def test_sumcheck_rounds_on_the_host(f, np_):
    """sumcheck_host.hpp: the rounds lurk_hip_sumcheck_prove_dev runs on the host once the tables are short (and the arithmetic of the
    round kernel): from 32 elements down to 2, every round's evaluation sums at 0, 2 (, 3) and every bound table against the
    definition in Python integers - e_t = sum_i comb((1 - t) P[i] + t P[h + i]), P[i] <- P[i] + r (P[h + i] - P[i])."""
    import ctypes

    L = H.lib()
    L.hh_sumcheck_round.restype = ctypes.c_size_t
    p = R.modulus(f)
    n = 32
    tabs = [[R.uniform_fe(300 + f + k, i, p) for i in range(n)] for k in range(np_)]
    tabs[0][3] = 0
    tabs[1][n - 1] = p - 1
    comb = (lambda a, b, c, d: a * (b * c - d)) if np_ == 4 else (lambda a, b: a * b)
    buf = np.concatenate([C.to_mont(f, C.ints_to_limbs(t)) for t in tabs]).reshape(np_, n, 4).copy()
    r = None
    length = n
    for rnd in range(6):
        if r is not None:   # what the round is expected to do first: bind the top variable to r
            h = length // 2
            tabs = [[(t[i] + r * (t[h + i] - t[i])) % p for i in range(h)] for t in tabs]
        h = len(tabs[0]) // 2
        want = [sum(comb(*[((1 - t) * tab[i] + t * tab[h + i]) % p for tab in tabs]) for i in range(h)) % p for t in ((0, 2, 3) if np_ == 4 else (0, 2))]
        ev = np.zeros((3, 4), dtype=np.uint64)
        cur = np.ascontiguousarray(buf[:, :length, :]).copy()
        r_l = None if r is None else C.to_mont(f, C.ints_to_limbs([r]))
        new_len = L.hh_sumcheck_round(f, np_, vp(cur), ctypes.c_size_t(length), None if r is None else vp(r_l), vp(ev))
        assert new_len == len(tabs[0]), rnd
        got_tabs = [C.limbs_to_ints(C.from_mont(f, np.ascontiguousarray(cur[k, :new_len, :]))) for k in range(np_)]
        assert got_tabs == tabs, rnd
        assert C.limbs_to_ints(C.from_mont(f, ev))[: len(want)] == want, rnd
        buf = np.ascontiguousarray(cur[:, :new_len, :]).copy()
        length = new_len
        if length < 2:
            break
        r = R.uniform_fe(310 + f, rnd, p)
    assert length == 1   # the sixth call bound the last variable: nothing left to sum, the tables are the final evaluations


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_pair_normalised_with_one_inversion(cn, c):
    """xyzz_pair_to_affine (the opening argument's L and R leave through one field inversion) = xyzz_to_affine of each, for two
    ordinary sums, with either or both the identity, and against the oracle's group law."""
    L = H.lib()
    B = C.synth_bases(c, 8)
    pts = C.affine_to_ints(c, B)
    z = np.zeros(0, dtype=np.uint32)
    cases = [((0, 5), (5, 8)), ((0, 3), (0, 0)), ((0, 0), (2, 6)), ((0, 0), (0, 0)), ((1, 2), (1, 2))]
    for (a0, a1), (b0, b1) in cases:
        out = np.zeros((4, 8), dtype=np.uint64)
        sa, sb = np.zeros(max(a1 - a0, 1), dtype=np.uint32), np.ones(max(b1 - b0, 1), dtype=np.uint32)
        L.hh_pair_to_affine(c, vp(B[a0:max(a1, a0 + 1)].copy()), vp(sa), ctypes.c_size_t(a1 - a0), vp(B[b0:max(b1, b0 + 1)].copy()), vp(sb), ctypes.c_size_t(b1 - b0), vp(out))
        got = C.affine_to_ints(c, out)
        assert got[0] == got[2] and got[1] == got[3], ((a0, a1), (b0, b1))
        wa = wb = None
        for pt in pts[a0:a1]:
            wa = R.ec_add(cn, wa, pt)
        for pt in pts[b0:b1]:
            wb = R.ec_add(cn, wb, R.ec_neg(cn, pt))
        assert got[0] == (wa or (0, 0)) and got[1] == (wb or (0, 0))


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_window_table_row_on_the_radix29_layer(cn, c):
    """msm_precompute_point (the body of msm_precompute_kernel: a chain of xyzz29_dbl, f29_invert, Montgomery's trick) with the bound
    assertions on: row w of a point's table is 2^(c w) P for the three shapes the library builds (16 x 16, 20 x 13, 8 x 32), the
    identity stays the identity, and f29_invert = fe_inv."""
    L = H.lib()
    assert L.hh_f29_checks_active() == 1
    B = C.synth_bases(c, 3)
    pts = C.affine_to_ints(c, B)
    for cw, W in ((16, 16), (20, 13), (8, 32), (6, 43)):
        for i in range(2 if cw in (16, 20) else 1):
            out = np.zeros((W, 8), dtype=np.uint64)
            L.hh_msm_precompute_row(c, vp(B[i:i + 1].copy()), cw, W, vp(out))
            got = C.affine_to_ints(c, out)
            want = pts[i]
            for w in range(W):
                assert got[w] == want, (cw, w)
                if w + 1 < W:
                    want = R.ec_mul(cn, 1 << cw, want)
    ident = np.zeros((1, 8), dtype=np.uint64)
    out = np.ones((16, 8), dtype=np.uint64)
    L.hh_msm_precompute_row(c, vp(ident), 16, 16, vp(out))
    assert not out.any()
    f = c  # the curve's base field: Pallas points have F_p coordinates (field id 0), Vesta points F_q (1)
    q = R.modulus(f)
    vals = [1, 2, q - 1, (1 << 254) - 1, 12345] + [R.uniform_fe(77, k, q) for k in range(6)]
    for v in vals:
        a = C.to_mont(f, C.ints_to_limbs([v]))
        o29, o32 = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        L.hh_f29_invert(f, vp(a), vp(o29), vp(o32))
        assert (o29 == o32).all(), hex(v)
        assert C.limbs_to_ints(C.from_mont(f, o29.reshape(1, 4)))[0] == pow(v, -1, q)


@pytest.mark.parametrize("kb,Wt", [(20, 13), (16, 16)])
@pytest.mark.parametrize("log_t,m", [(0, 1 << 17), (1, 1 << 16), (4, 1 << 16), (4, 64), (8, 16), (12, 4)])
def test_key_fold_plan_digits_recombine_to_the_weights(kb, Wt, log_t, m):
    """keyfold_plan.hpp (the host half of lurk_hip_msm_ctx_fold_key_dev): for uniform weights and the edge values (0, 1, 2^k around the
    slot and window boundaries, q - 1, 2^255 - 1) the signed sub-digits read back from the sorted lists the kernel walks recombine to
    the weight, every digit is within its slot's range, every (weight, window, slot) is listed at most once, and the window groups grow
    when there are too few outputs to fill the chip; a weight whose digits carry out of the windows is refused (the library only passes
    canonical field elements, below 2^255)."""
    import ctypes

    lib = H.lib()
    T = 1 << log_t
    q = R.modulus(1)
    special = [0, 1, 2, q - 1, (1 << 255) - 1, 1 << 254, (1 << 20) - 1, 1 << 19, (1 << 19) + 1, (1 << 16) - 1, 1 << 15, (1 << 240) + (1 << 239), (1 << 7) + (1 << 6)]
    ws = [special[b] if b < len(special) else R.uniform_fe(190, b, q) for b in range(T)]
    if T == 1:
        ws = [R.uniform_fe(191, 0, q)]
    arr = np.zeros((T, 4), dtype=np.uint64)
    for b, w in enumerate(ws):
        for k in range(4):
            arr[b, k] = (w >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    U, groups, maxmag = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    off, wd = (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
    digits = np.zeros(T * Wt * 8, dtype=np.int32)
    rc = lib.hh_keyfold_plan(arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(T), ctypes.c_size_t(m), kb, Wt, ctypes.byref(U), ctypes.byref(groups),
                             ctypes.byref(maxmag), off, wd, digits.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    U, groups, maxmag = U.value, groups.value, maxmag.value
    assert 1 <= U <= 8 and sum(wd[u] for u in range(U)) == kb and [off[u] for u in range(U)] == [sum(wd[v] for v in range(u)) for u in range(U)]
    assert maxmag == 1 << (max(wd[u] for u in range(U)) - 1)
    assert groups == 1 if m * U >= 1 << 17 else groups == min(Wt, -(-(1 << 17) // (m * U)))
    d = digits[: T * Wt * U].reshape(T, Wt, U)
    for b, w in enumerate(ws):
        got = sum(int(d[b, j, u]) << (j * kb + off[u]) for j in range(Wt) for u in range(U))
        assert got == w, (b, hex(w))
        assert all(abs(int(d[b, j, u])) <= 1 << (wd[u] - 1) for j in range(Wt) for u in range(U))
    arr[0] = [0xFFFFFFFFFFFFFFFF] * 4  # 2^256 - 1: its signed digits carry out of a 256-bit window set (16 x 16), 13 x 20 bits hold it
    rc = lib.hh_keyfold_plan(arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(T), ctypes.c_size_t(m), kb, Wt, ctypes.byref(ctypes.c_int()),
                             ctypes.byref(ctypes.c_int()), ctypes.byref(ctypes.c_int()), off, wd, digits.ctypes.data_as(ctypes.c_void_p))
    assert rc == (1 if kb * Wt == 256 else 0)


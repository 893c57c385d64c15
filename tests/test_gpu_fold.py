"""GPU parity for the device-resident fold through the C ABI: multiply_vec, cross term and
fold against the oracle on synthetic R1CS shapes, plus the size-independent folding identity at step-circuit
size (the folded (z, E) satisfies the relaxed instance).  Parity unpinned upstream (no vectors exist)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def _shape(f, A, B, Cm, m, nv, nio):
    from lurk_beta_amd import R1CSShape

    mont = lambda M: (M[0], M[1], C.to_mont(f, M[2]))
    return R1CSShape(f, m, nv, nio, mont(A), mont(B), mont(Cm))


@pytest.mark.parametrize("f", [0, 1, 2])
def test_multiply_vec_cross_term_fold_match_oracle(hip, f):
    from lurk_beta_amd import fold_vec

    p = R.modulus(f)
    m, nv, nio = 3000, 2500, 2
    A, B, Cm, z2 = C.synth_r1cs(f, m, nv, nio, seed=11)
    sh = _shape(f, A, B, Cm, m, nv, nio)
    info = sh.info()
    assert info["nnz"] == (A[1].size, B[1].size, Cm[1].size) and info["distinct_coefficients"] <= 32 + m
    z1 = C.synth_scalars(f, 8, 0, nv + 1 + nio)
    d_z1, d_z2 = _dev(C.to_mont(f, z1)), _dev(C.to_mont(f, z2))
    got = [C.from_mont(f, _host(x)) for x in sh.multiply_vec(d_z1)]
    want = [C.spmv(f, *M, z1) for M in (A, B, Cm)]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    u1 = C.limbs_to_ints(z1[nv:nv + 1])[0]
    az2, bz2, cz2 = (C.spmv(f, *M, z2) for M in (A, B, Cm))
    t_want = C.cross_term(f, *want, az2, bz2, cz2, u1, 1)
    d_t = sh.cross_term(d_z1, d_z2)
    assert np.array_equal(C.from_mont(f, _host(d_t)), t_want)
    r = R.uniform_fe(72, f, p)
    r_mont = C.to_mont(f, C.ints_to_limbs([r]))
    assert np.array_equal(C.from_mont(f, _host(fold_vec(f, d_z1, d_z2, r_mont))), C.axpy(f, z1, z2, r))
    sh.close()


def test_edge_shapes(hip):
    """Empty rows, a row longer than the 64-term re-entry period, zero coefficients, all-(p-1) operands, empty shape."""
    from lurk_beta_amd import R1CSShape, fold_vec

    f = 1
    p = R.modulus(f)
    ncols = 700
    rng = np.random.default_rng(2)
    lens = np.array([0, 1, 700, 0, 65, 64, 129, 5], dtype=np.uint64)
    indptr = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = rng.integers(0, ncols, nnz).astype(np.uint64)
    vals = [p - 1] * nnz
    for k in range(0, nnz, 7):
        vals[k] = 0
    for k in range(3, nnz, 5):
        vals[k] = R.uniform_fe(73, k, p)
    data = C.ints_to_limbs(vals)
    M = (indptr, indices, data)
    z = C.ints_to_limbs([p - 1] * ncols)
    z[::3] = C.synth_scalars(f, 9, 0, ncols)[::3]
    nv = ncols - 3
    sh = R1CSShape(f, len(lens), nv, 2, *[(indptr, indices, C.to_mont(f, data))] * 3)
    got = [C.from_mont(f, _host(x)) for x in sh.multiply_vec(_dev(C.to_mont(f, z)))]
    want = C.spmv(f, *M, z)
    assert all(np.array_equal(g, want) for g in got)
    sh.close()
    empty = (np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))
    sh0 = R1CSShape(f, 0, 4, 1, empty, empty, empty)
    assert sh0.info()["nnz"] == (0, 0, 0)
    sh0.close()
    import torch

    e = torch.empty((0, 4), dtype=torch.int64, device="cuda")
    assert fold_vec(f, e, e, C.ints_to_limbs([5])).shape[0] == 0


def test_bad_arguments(hip):
    from lurk_beta_amd import LurkHipError, R1CSShape

    ip = np.array([0, 1], dtype=np.uint64)
    with pytest.raises(LurkHipError):  # column out of range
        R1CSShape(1, 1, 2, 1, *[(ip, np.array([9], dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))] * 3)
    with pytest.raises(LurkHipError):  # indptr not starting at 0
        R1CSShape(1, 1, 2, 1, *[(np.array([1, 1], dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))] * 3)


def test_cross_term_two_streams_one_shape(hip):
    """Calls on one shape keep no state outside their arguments: two cross terms with different (z1, z2) enqueued
    back to back on two streams, several rounds, every result exact."""
    import torch

    f, m, nv, nio = 1, 60000, 50000, 2
    A, B, Cm, z2 = C.synth_r1cs(f, m, nv, nio, seed=17)
    sh = _shape(f, A, B, Cm, m, nv, nio)
    zs = [C.synth_scalars(f, 80 + k, k % 2, nv + 1 + nio) for k in range(4)]
    mats = [[C.spmv(f, *M, z) for M in (A, B, Cm)] for z in zs]
    us = [C.limbs_to_ints(z[nv:nv + 1])[0] for z in zs]
    pairs = [(0, 1), (2, 3)]
    want = [C.cross_term(f, *mats[a], *mats[b], us[a], us[b]) for a, b in pairs]
    d = [_dev(C.to_mont(f, z)) for z in zs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty((m, 4), dtype=torch.int64, device="cuda") for _ in pairs]
    torch.cuda.synchronize()
    for _ in range(20):
        for k, (a, b) in enumerate(pairs):
            sh.cross_term(d[a], d[b], out=outs[k], stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        assert np.array_equal(C.from_mont(f, _host(outs[k])), want[k])
    sh.close()


@pytest.mark.parametrize("rc", [100, 900])
def test_folding_identity_at_step_circuit_size(hip, rc):
    """Step-circuit sizes of BASELINE configs[0] and [3] (rc = 100: ~1.11 M constraints, ~0.91 M variables; rc = 900:
    ~10.0 M / ~8.2 M): fold a relaxed instance with a strictly satisfied one on the GPU, then check
    (Az o Bz) = u Cz + E row by row."""
    from lurk_beta_amd import fold_vec

    f, m, nv, nio = 1, 11141 * rc, 9119 * rc, 2
    p = R.modulus(f)
    A, B, Cm, z2 = C.synth_r1cs(f, m, nv, nio, seed=13)
    sh = _shape(f, A, B, Cm, m, nv, nio)
    z1 = C.synth_scalars(f, 8, 0, nv + 1 + nio)
    u1 = C.limbs_to_ints(z1[nv:nv + 1])[0]
    d_z1, d_z2 = _dev(C.to_mont(f, z1)), _dev(C.to_mont(f, z2))
    zero = np.zeros((m, 4), dtype=np.uint64)
    az1, bz1, cz1 = [C.from_mont(f, _host(x)) for x in sh.multiply_vec(d_z1)]
    e1 = C.relaxed_residual(f, az1, bz1, cz1, u1, zero)       # E1 := Az1 o Bz1 - u1 Cz1 makes (z1, E1) a relaxed witness
    d_t = sh.cross_term(d_z1, d_z2)
    r = R.uniform_fe(74, 0, p)
    r_mont = C.to_mont(f, C.ints_to_limbs([r]))
    d_z = fold_vec(f, d_z1, d_z2, r_mont)
    d_e = fold_vec(f, _dev(C.to_mont(f, e1)), d_t, r_mont)
    az, bz, cz = [C.from_mont(f, _host(x)) for x in sh.multiply_vec(d_z)]
    u = C.limbs_to_ints(C.from_mont(f, _host(d_z[nv:nv + 1])))[0]
    assert u == (u1 + r) % p
    assert not C.relaxed_residual(f, az, bz, cz, u, C.from_mont(f, _host(d_e))).any()
    sh.close()


@pytest.mark.parametrize("f", [0, 1, 2])
def test_cached_products_cross_term_and_multi_fold_match_oracle(hip, f):
    """Round 6: lurk_hip_r1cs_cross_term_cached_dev (T from the cached A z1, B z1, C z1 and the gathers of z2 alone; long rows through
    the lanes-per-row path) and lurk_hip_fold_vecs_dev (several folds in one launch, in place) against the oracle, and the linearity
    the cache rests on: folding the cached products with r gives the products of the folded z."""
    from lurk_beta_amd import fold_vecs

    p = R.modulus(f)
    m, nv, nio = 3000, 2500, 2
    A, B, Cm, z2 = C.synth_r1cs(f, m, nv, nio, seed=21)
    # make a few rows long (more than FOLD_LONG = 32 entries, one beyond the 64-term re-entry period) so that both row paths run
    rng = np.random.default_rng(5)
    ip, ix, dv = A
    cnt = np.diff(ip.astype(np.int64))
    extra_rows = [7, 1500, 2999]
    new_ip, new_ix, new_dv = [0], [], []
    for i in range(m):
        lo, hi = int(ip[i]), int(ip[i + 1])
        cols, vals = list(ix[lo:hi]), list(dv[lo:hi])
        if i in extra_rows:
            k = 40 if i != 1500 else 200
            cols += list(rng.integers(0, nv + 1 + nio, k).astype(np.uint64))
            vals += list(C.synth_scalars(f, 300 + i, 0, k))
        new_ix += cols
        new_dv += vals
        new_ip.append(len(new_ix))
    A = (np.array(new_ip, dtype=np.uint64), np.array(new_ix, dtype=np.uint64), np.array(new_dv, dtype=np.uint64).reshape(-1, 4))
    sh = _shape(f, A, B, Cm, m, nv, nio)
    z1 = C.synth_scalars(f, 18, 0, nv + 1 + nio)
    d_z1, d_z2 = _dev(C.to_mont(f, z1)), _dev(C.to_mont(f, z2))
    abc1 = sh.multiply_vec(d_z1)
    want1 = [C.spmv(f, *M, z1) for M in (A, B, Cm)]
    want2 = [C.spmv(f, *M, z2) for M in (A, B, Cm)]
    u1 = C.limbs_to_ints(z1[nv:nv + 1])[0]
    u2 = C.limbs_to_ints(z2[nv:nv + 1])[0]
    t_want = C.cross_term(f, *want1, *want2, u1, u2)
    u1_mont = C.to_mont(f, z1[nv:nv + 1])
    d_t, abc2 = sh.cross_term_cached(d_z2, abc1, u1_mont)
    assert np.array_equal(C.from_mont(f, _host(d_t)), t_want)
    assert np.array_equal(_host(d_t), _host(sh.cross_term(d_z1, d_z2)))  # bit for bit the six-gather kernel's T
    for g, w in zip(abc2, want2):
        assert np.array_equal(C.from_mont(f, _host(g)), w)
    # finish(r): [z, E, A z1, B z1, C z1] fold in one launch; the three products in place
    r = R.uniform_fe(73, f, p)
    r_mont = C.to_mont(f, C.ints_to_limbs([r]))
    e1 = C.synth_scalars(f, 19, 0, m)
    d_e1 = _dev(C.to_mont(f, e1))
    outs = fold_vecs(f, [(d_z1, d_z2), (d_e1, d_t)] + list(zip(abc1, abc2)), r_mont)
    zf, ef = C.axpy(f, z1, z2, r), C.axpy(f, e1, t_want, r)
    assert np.array_equal(C.from_mont(f, _host(outs[0])), zf) and np.array_equal(C.from_mont(f, _host(outs[1])), ef)
    import torch

    inplace = [x.clone() for x in abc1]
    fold_vecs(f, list(zip(inplace, abc2)), r_mont, outs=inplace)
    torch.cuda.synchronize()
    for g, M in zip(inplace, (A, B, Cm)):
        assert np.array_equal(C.from_mont(f, _host(g)), C.spmv(f, *M, zf))  # A (z1 + r z2) = A z1 + r A z2
    # the cache's own fold rides in the cross-term launch: given the previous step's products (here: z2's) and its challenge, the cached
    # rows become A (z1 + r z2), B (..), C (..) IN PLACE and T is the cross term of (z1 + r z2, z3)
    z3 = np.concatenate([C.synth_scalars(f, 31, 1, nv), C.ints_to_limbs([1]), C.synth_scalars(f, 32, 0, nio)])
    d_z3 = _dev(C.to_mont(f, z3))
    fused = [x.clone() for x in abc1]
    uf = C.to_mont(f, zf[nv:nv + 1])  # the running u after the fold (the host folds scalars itself)
    t3, abc3 = sh.cross_term_cached(d_z3, fused, uf, prev=abc2, r_prev_mont=r_mont)
    torch.cuda.synchronize()
    for g, M in zip(fused, (A, B, Cm)):
        assert np.array_equal(C.from_mont(f, _host(g)), C.spmv(f, *M, zf))
    want3 = [C.spmv(f, *M, z3) for M in (A, B, Cm)]
    t3_want = C.cross_term(f, *[C.spmv(f, *M, zf) for M in (A, B, Cm)], *want3, C.limbs_to_ints(zf[nv:nv + 1])[0], 1)
    assert np.array_equal(C.from_mont(f, _host(t3)), t3_want)
    for g, w in zip(abc3, want3):
        assert np.array_equal(C.from_mont(f, _host(g)), w)
    from lurk_beta_amd import LurkHipError

    with pytest.raises(LurkHipError, match="go together"):
        sh.cross_term_cached(d_z3, fused, uf, prev=abc2)  # products without their challenge
    # edge: zero vectors in the list, a single vector, the empty call
    assert fold_vecs(f, [], r_mont) == []
    one = fold_vecs(f, [(d_e1, d_t)], r_mont)
    assert np.array_equal(C.from_mont(f, _host(one[0])), ef)
    with pytest.raises(LurkHipError):
        fold_vecs(f, [(d_e1, d_t)] * 9, r_mont)
    sh.close()

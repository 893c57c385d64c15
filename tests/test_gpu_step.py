"""M1: consecutive folding steps on BOTH curves of the cycle through lurk_hip_fold_step_{begin,finish}; after every step the
running pair must satisfy the relaxed instance (A z o B z = u C z + E row by row, arecibo's is_sat_relaxed), its commitments must
be the Pedersen commitments of the folded vectors (RecursiveSNARK::verify re-commits W and E), and every
value must equal the oracle's fold.  Reference loop: /root/reference/src/proof/nova.rs:282-295.  Parity unpinned upstream."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu


def _product_shape(f, m, nfree, nio, seed):
    """A, B random sparse over the free variables, u and X; C picks the row's own product variable: any free assignment
    extends to a satisfying witness, so one shape takes a different fresh instance every step."""
    rng = np.random.default_rng(seed)
    p = R.modulus(f)
    nv = nfree + m
    coeffs = C.ints_to_limbs([1, p - 1, 2, 3, p - 5, 1 << 20, R.uniform_fe(95, seed, p)])

    def rand_mat():
        cnt = rng.integers(1, 5, m).astype(np.uint64)
        cnt[rng.integers(0, m, max(1, m // 200))] = 70  # a few long rows (the wave-per-row kernel)
        indptr = np.zeros(m + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        cols = rng.integers(0, nfree + 1 + nio, nnz)
        cols = np.where(cols >= nfree, cols + m, cols).astype(np.uint64)  # skip the product block: [free | products | u | X]
        return indptr, cols, np.ascontiguousarray(coeffs[rng.integers(0, len(coeffs), nnz)])

    A, B = rand_mat(), rand_mat()
    Cm = (np.arange(m + 1, dtype=np.uint64), (nfree + np.arange(m)).astype(np.uint64), np.tile(C.ints_to_limbs([1]), (m, 1)))
    return A, B, Cm, nv


def _fresh(f, A, B, m, nfree, nio, stream_id):
    """[W2 | 1 | X2] strictly satisfying the product shape."""
    p = R.modulus(f)
    free = C.synth_scalars(f, stream_id, 1, nfree)
    x2 = C.synth_scalars(f, stream_id + 1, 0, nio)
    z = np.concatenate([free, np.zeros((m, 4), dtype=np.uint64), C.ints_to_limbs([1]), x2])
    az, bz = C.limbs_to_ints(C.spmv(f, *A, z)), C.limbs_to_ints(C.spmv(f, *B, z))
    z[nfree:nfree + m] = C.ints_to_limbs([a * b % p for a, b in zip(az, bz)])
    return z, x2


@pytest.mark.parametrize("curve,m,nfree", [(0, 20000, 9000), (1, 3000, 1500)])
def test_three_consecutive_steps(hip, curve, m, nfree):
    import torch

    from lurk_beta_amd import CommitmentKey, FoldingContext, R1CSShape, point_to_affine, public_io

    f = 1 if curve == 0 else 0  # scalar field of the curve
    p = R.modulus(f)
    nio = 6 if curve == 0 else 2  # Z1: a Lurk step's IO is [tag, hash] x (expr, env, cont)
    assert public_io([(1, 11), (2, 22), (0x1000, 33)]) == [1, 11, 2, 22, 0x1000, 33]
    A, B, Cm, nv = _product_shape(f, m, nfree, nio, seed=3 + curve)
    mont = lambda M: (M[0], M[1], C.to_mont(f, M[2]))
    shape = R1CSShape(f, m, nv, nio, mont(A), mont(B), mont(Cm))
    bases = C.synth_bases(curve, max(m, nv))
    key = CommitmentKey(curve, bases, precompute=bool(curve))
    ctx = FoldingContext(curve, shape, key)
    commit = lambda v: C.jac_to_affine(curve, C.msm_pippenger(curve, bases[: len(v)], v))
    # oracle-side running pair: the default relaxed instance
    z1 = np.zeros((nv + 1 + nio, 4), dtype=np.uint64)
    e1 = np.zeros((m, 4), dtype=np.uint64)
    for step in range(3):
        z2, x2 = _fresh(f, A, B, m, nfree, nio, 100 + 10 * step + curve)
        w2_mont = C.to_mont(f, z2[:nv])
        w2_arg = torch.from_numpy(w2_mont.view(np.int64)).cuda() if step == 1 else w2_mont  # device-resident W2 on one step
        cw, ct = ctx.begin(w2_arg, C.to_mont(f, x2), stream=torch.cuda.current_stream().cuda_stream if step == 1 else None)
        m1 = [C.spmv(f, *M, z1) for M in (A, B, Cm)]
        m2 = [C.spmv(f, *M, z2) for M in (A, B, Cm)]
        u1 = C.limbs_to_ints(z1[nv:nv + 1])[0]
        t = C.cross_term(f, *m1, *m2, u1, 1)
        assert point_to_affine(curve, cw) == commit(z2[:nv])
        assert point_to_affine(curve, ct) == commit(t)
        r = R.uniform_fe(96, step * 2 + curve, p) >> 128  # arecibo squeezes NUM_CHALLENGE_BITS = 128 bits
        ctx.finish(C.to_mont(f, C.ints_to_limbs([r])))
        z1, e1 = C.axpy(f, z1, z2, r), C.axpy(f, e1, t, r)
        gz, ge = ctx.read()
        assert np.array_equal(C.from_mont(f, gz), z1) and np.array_equal(C.from_mont(f, ge), e1)
        u = C.limbs_to_ints(z1[nv:nv + 1])[0]
        assert u == (u1 + r) % p
        az, bz, cz = [C.spmv(f, *M, z1) for M in (A, B, Cm)]
        assert not C.relaxed_residual(f, az, bz, cz, u, e1).any()           # is_sat_relaxed
        assert point_to_affine(curve, ctx.comm_W) == commit(z1[:nv])          # comm_W1 + r comm_W2 = commit(W1 + r W2)
        assert point_to_affine(curve, ctx.comm_E) == commit(e1)               # comm_E1 + r comm_T  = commit(E1 + r T)
    from lurk_beta_amd import LurkHipError

    with pytest.raises(LurkHipError):
        ctx.finish(C.ints_to_limbs([1]))  # no step open
    ctx.begin(C.to_mont(f, z2[:nv]), C.to_mont(f, x2))
    with pytest.raises(LurkHipError):
        ctx.begin(C.to_mont(f, z2[:nv]), C.to_mont(f, x2))  # a step is already open
    ctx.close()
    key.close()
    shape.close()


@pytest.mark.parametrize("curve,m,nfree", [(0, 20000, 9000), (1, 3000, 1500)])
def test_steps_with_instances_staged_ahead(hip, curve, m, nfree):
    """lurk_hip_fold_step_prefetch / begin_prefetched: the step circuit's range of W2 is staged (and its commitment started) one
    step ahead, the augmented circuit's ranges around it arrive with begin.  Every step must give what the plain begin gives:
    comm_W2 = commit(whole W2) by linearity, same T, same folded pair."""
    import torch

    from lurk_beta_amd import CommitmentKey, FoldingContext, LurkHipError, R1CSShape, point_to_affine

    f = 1 if curve == 0 else 0
    p = R.modulus(f)
    nio = 6 if curve == 0 else 2
    A, B, Cm, nv = _product_shape(f, m, nfree, nio, seed=13 + curve)
    mont = lambda M: (M[0], M[1], C.to_mont(f, M[2]))
    shape = R1CSShape(f, m, nv, nio, mont(A), mont(B), mont(Cm))
    bases = C.synth_bases(curve, max(m, nv))
    key = CommitmentKey(curve, bases, precompute=bool(curve))
    key.reserve(max(m, nv), 4)
    ctx = FoldingContext(curve, shape, key)
    commit = lambda v: C.jac_to_affine(curve, C.msm_pippenger(curve, bases[: len(v)], v))
    lo, hi = nv // 50, nv - nv // 30  # the staged body is [lo, hi); prefix and suffix arrive late
    steps = 4
    fresh = [_fresh(f, A, B, m, nfree, nio, 300 + 10 * k + curve) for k in range(steps)]
    w2m = [C.to_mont(f, z2[:nv]) for z2, _ in fresh]

    def stage(k):
        body = np.ascontiguousarray(w2m[k][lo:hi])
        if k % 2:  # device-resident range on odd steps
            ctx.prefetch(torch.from_numpy(body.view(np.int64)).cuda(), lo, stream=torch.cuda.current_stream().cuda_stream)
        else:
            ctx.prefetch(body, lo)

    z1 = np.zeros((nv + 1 + nio, 4), dtype=np.uint64)
    e1 = np.zeros((m, 4), dtype=np.uint64)
    stage(0)
    for k in range(steps):
        if k + 1 < steps:
            stage(k + 1)  # two instances staged: k (about to open) and k + 1
        if k == 0:
            with pytest.raises(LurkHipError):
                ctx.prefetch(w2m[0][lo:hi], lo)  # a third one is refused
            with pytest.raises(LurkHipError):
                ctx.begin(w2m[0], C.to_mont(f, fresh[0][1]))  # the plain form is refused while instances are staged
        z2, x2 = fresh[k]
        patches = [(0, w2m[k][:lo]), (hi, w2m[k][hi:])] if k != 2 else [(hi, w2m[k][hi:]), (0, w2m[k][:lo]), (5, w2m[k][5:5])]
        cw, ct = ctx.begin_prefetched(C.to_mont(f, x2), patches)
        m1 = [C.spmv(f, *M, z1) for M in (A, B, Cm)]
        m2 = [C.spmv(f, *M, z2) for M in (A, B, Cm)]
        u1 = C.limbs_to_ints(z1[nv:nv + 1])[0]
        t = C.cross_term(f, *m1, *m2, u1, 1)
        assert point_to_affine(curve, cw) == commit(z2[:nv])
        assert point_to_affine(curve, ct) == commit(t)
        r = R.uniform_fe(97, k * 2 + curve, p) >> 128
        ctx.finish(C.to_mont(f, C.ints_to_limbs([r])))
        z1, e1 = C.axpy(f, z1, z2, r), C.axpy(f, e1, t, r)
        gz, ge = ctx.read()
        assert np.array_equal(C.from_mont(f, gz), z1) and np.array_equal(C.from_mont(f, ge), e1)
        assert point_to_affine(curve, ctx.comm_W) == commit(z1[:nv])
        assert point_to_affine(curve, ctx.comm_E) == commit(e1)
    # nothing staged any more: the plain form works again, and a whole witness staged ahead needs no patches
    with pytest.raises(LurkHipError):
        ctx.begin_prefetched(C.to_mont(f, fresh[0][1]))
    z2, x2 = fresh[1]
    ctx.prefetch(w2m[1])
    cw, ct = ctx.begin_prefetched(C.to_mont(f, x2))
    assert point_to_affine(curve, cw) == commit(z2[:nv])
    ctx.finish(C.to_mont(f, C.ints_to_limbs([3])))
    cw2, _ = ctx.begin(w2m[1], C.to_mont(f, x2))
    assert point_to_affine(curve, cw2) == point_to_affine(curve, cw)
    ctx.close()
    key.close()
    shape.close()


def test_shape_and_key_must_match_the_curve(hip):
    from lurk_beta_amd import CommitmentKey, FoldingContext, LurkHipError, R1CSShape

    empty = (np.zeros(2, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))
    shape = R1CSShape(0, 1, 2, 1, empty, empty, empty)  # over Fp: Vesta's scalar field
    key = CommitmentKey(0, C.synth_bases(0, 4))
    with pytest.raises(LurkHipError):
        FoldingContext(0, shape, key)  # Pallas needs a shape over Fq
    key.close()
    shape.close()

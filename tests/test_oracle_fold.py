"""CPU: the C oracle's relaxed-R1CS folding functions against the Python-int restatement, and the folding identity
itself (the folded pair satisfies the relaxed instance) on a synthetic shape."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R


@pytest.mark.parametrize("f", [0, 1, 2])
def test_c_oracle_matches_python_ints(f):
    p = R.modulus(f)
    A, B, Cm, z2 = C.synth_r1cs(f, 200, 150, 2, seed=3)
    z = C.limbs_to_ints(z2)
    for M in (A, B, Cm):
        got = C.limbs_to_ints(C.spmv(f, *M, z2))
        want = R.spmv(p, [int(x) for x in M[0]], [int(x) for x in M[1]], C.limbs_to_ints(M[2]), z)
        assert got == want
    v = [C.synth_scalars(f, 10 + i, 0, 50) for i in range(6)]
    u1, u2, r = R.uniform_fe(70, 0, p), 1, R.uniform_fe(70, 1, p)
    assert C.limbs_to_ints(C.cross_term(f, *v, u1, u2)) == R.cross_term(p, *[C.limbs_to_ints(x) for x in v], u1, u2)
    assert C.limbs_to_ints(C.axpy(f, v[0], v[1], r)) == R.axpy(p, C.limbs_to_ints(v[0]), C.limbs_to_ints(v[1]), r)


def test_fold_of_relaxed_and_strict_instance_is_satisfied():
    f, m, nv, nio = 1, 3000, 2500, 2
    p = R.modulus(f)
    A, B, Cm, z2 = C.synth_r1cs(f, m, nv, nio, seed=5)
    zero = np.zeros((m, 4), dtype=np.uint64)
    # the fresh instance is strictly satisfied (u = 1, E = 0)
    assert not C.relaxed_residual(f, C.spmv(f, *A, z2), C.spmv(f, *B, z2), C.spmv(f, *Cm, z2), 1, zero).any()
    # any z1 with E1 := Az1 o Bz1 - u1 Cz1 is a relaxed witness
    z1 = C.synth_scalars(f, 8, 0, nv + 1 + nio)
    u1 = C.limbs_to_ints(z1[nv:nv + 1])[0]
    az1, bz1, cz1 = (C.spmv(f, *M, z1) for M in (A, B, Cm))
    e1 = C.relaxed_residual(f, az1, bz1, cz1, u1, zero)
    az2, bz2, cz2 = (C.spmv(f, *M, z2) for M in (A, B, Cm))
    t = C.cross_term(f, az1, bz1, cz1, az2, bz2, cz2, u1, 1)
    r = R.uniform_fe(71, 0, p)
    z = C.axpy(f, z1, z2, r)
    e = C.axpy(f, e1, t, r)
    u = C.limbs_to_ints(z[nv:nv + 1])[0]
    assert u == (u1 + r) % p
    assert not C.relaxed_residual(f, C.spmv(f, *A, z), C.spmv(f, *B, z), C.spmv(f, *Cm, z), u, e).any()

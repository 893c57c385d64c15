"""CPU: the Spartan-style SNARK oracle (oracle/spartan_ref.py) is complete and sound on small relaxed R1CS instances: honest
proofs verify (strict and folded instances), tampered proofs and wrong statements do not."""
import copy
import random

import pytest

from oracle import pyref as R
from oracle import spartan_ref as S


def product_instance(curve, num_cons, num_vars, nio, seed, folded):
    """Rows: (random combination of free vars, u, X) * (another) = own product variable.  Returns mats, X, u, W, E."""
    q = R.CURVES[curve]["order"]
    rng = random.Random(seed)
    nfree = num_vars - num_cons
    assert nfree >= 1 and 1 + nio <= num_vars
    cols = list(range(nfree)) + [num_vars + k for k in range(1 + nio)]  # free vars, u, X

    def rand_mat():
        indptr, indices, data = [0], [], []
        for _ in range(num_cons):
            for c in rng.sample(cols, rng.randint(1, min(3, len(cols)))):
                indices.append(c)
                data.append(rng.choice([1, q - 1, 2, rng.randrange(q)]))
            indptr.append(len(indices))
        return indptr, indices, data

    A, B = rand_mat(), rand_mat()
    Cm = (list(range(num_cons + 1)), [nfree + i for i in range(num_cons)], [1] * num_cons)
    mats = (A, B, Cm)

    def fresh(s):
        r2 = random.Random(s)
        free = [r2.randrange(q) for _ in range(nfree)]
        X = [r2.randrange(q) for _ in range(nio)]
        z = free + [0] * num_cons + [1] + X + [0] * (2 * num_vars - num_vars - 1 - nio)
        az, bz, _ = S.matrices_times(q, mats, z)
        W = free + [a * b % q for a, b in zip(az, bz)]
        return W, X

    W2, X2 = fresh(seed + 1)
    if not folded:
        return mats, X2, 1, W2, [0] * num_cons
    W1, X1 = fresh(seed + 2)
    z1 = W1 + [1] + X1 + [0] * (num_vars - 1 - nio)
    z2 = W2 + [1] + X2 + [0] * (num_vars - 1 - nio)
    m1, m2 = S.matrices_times(q, mats, z1), S.matrices_times(q, mats, z2)
    T = R.cross_term(q, *m1, *m2, 1, 1)
    r = rng.randrange(q)
    W = [(a + r * b) % q for a, b in zip(W1, W2)]
    X = [(a + r * b) % q for a, b in zip(X1, X2)]
    return mats, X, (1 + r) % q, W, [r * t % q for t in T]


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
@pytest.mark.parametrize("num_cons,num_vars,folded", [(4, 8, False), (8, 16, True), (16, 32, True)])
def test_complete_and_sound(curve, num_cons, num_vars, folded):
    q = R.CURVES[curve]["order"]
    mats, X, u, W, E = product_instance(curve, num_cons, num_vars, 2, 5, folded)
    N = max(num_cons, num_vars)
    key = R.synth_bases(curve, N + 1)
    ck, ck_c = key[:N], key[N]
    comm_W, comm_E = R.msm_naive(curve, W, ck), R.msm_naive(curve, E, ck)
    proof = S.prove(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, W, E)
    assert S.verify(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, proof)
    # wrong statement / tampered proof
    assert not S.verify(curve, mats, num_cons, num_vars, [(X[0] + 1) % q] + X[1:], ck, ck_c, comm_W, comm_E, u, proof)
    assert not S.verify(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, (u + 1) % q, proof)
    for field in ("eval_W", "eval_E", "ipa_a"):
        bad = copy.deepcopy(proof)
        bad[field] = (bad[field] + 1) % q
        assert not S.verify(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, bad), field
    bad = copy.deepcopy(proof)
    bad["polys_inner"][1][0] = (bad["polys_inner"][1][0] + 1) % q
    assert not S.verify(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, bad)
    # an unsatisfied instance: the honest prover's proof is rejected (the outer claim is not 0)
    E2 = list(E)
    E2[0] = (E2[0] + 1) % q
    comm_E2 = R.msm_naive(curve, E2, ck)
    p2 = S.prove(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E2, u, W, E2)
    assert not S.verify(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E2, u, p2)

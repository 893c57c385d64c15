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


def _to_arrays(mats, X, W, E):
    import numpy as np

    from oracle import coracle as C

    m = [(np.array(ip, dtype=np.uint64), np.array(ix, dtype=np.uint64), C.ints_to_limbs(d) if len(d) else np.zeros((0, 4), dtype=np.uint64)) for ip, ix, d in mats]
    return m, C.ints_to_limbs(W), C.ints_to_limbs(E)


@pytest.mark.parametrize("curve_id,curve,num_cons,num_vars,folded",
                         [(0, "pallas", 4, 8, False), (1, "vesta", 8, 16, True), (0, "pallas", 16, 64, True), (1, "vesta", 16, 32, False)])
def test_fast_oracle_produces_the_reference_oracles_proof(curve_id, curve, num_cons, num_vars, folded):
    """oracle/spartan_fast.py (C vector work, msm_fast commitments: the oracle that runs at 2^14 .. 2^20) against
    oracle/spartan_ref.py (Python integers): the same proof element for element, and each verifier accepts the other's proof."""
    import numpy as np

    from oracle import coracle as C
    from oracle import spartan_fast as SF

    q = R.CURVES[curve]["order"]
    mats, X, u, W, E = product_instance(curve, num_cons, num_vars, 2, 7, folded)
    N = max(num_cons, num_vars)
    key = R.synth_bases(curve, N + 1)
    ck, ck_c = key[:N], key[N]
    key_arr = C.synth_bases(curve_id, N + 1)
    assert [None if pt == (0, 0) else pt for pt in C.affine_to_ints(curve_id, key_arr)] == key
    comm_W, comm_E = R.msm_naive(curve, W, ck), R.msm_naive(curve, E, ck)
    ref = S.prove(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, W, E)
    m_arr, W_arr, E_arr = _to_arrays(mats, X, W, E)
    fast = SF.prove(curve_id, m_arr, num_cons, num_vars, X, key_arr, comm_W, comm_E, u, W_arr, E_arr)
    assert fast == ref
    assert SF.verify(curve_id, m_arr, num_cons, num_vars, X, key_arr, comm_W, comm_E, u, ref)
    assert S.verify(curve, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, fast)
    assert not SF.verify(curve_id, m_arr, num_cons, num_vars, [(X[0] + 1) % q] + X[1:], key_arr, comm_W, comm_E, u, ref)
    for field in ("eval_W", "eval_E", "ipa_a"):
        bad = copy.deepcopy(ref)
        bad[field] = (bad[field] + 1) % q
        assert not SF.verify(curve_id, m_arr, num_cons, num_vars, X, key_arr, comm_W, comm_E, u, bad), field
    bad = copy.deepcopy(ref)
    bad["ipa_L"][0] = bad["ipa_R"][0]
    assert not SF.verify(curve_id, m_arr, num_cons, num_vars, X, key_arr, comm_W, comm_E, u, bad)
    bad = copy.deepcopy(ref)
    bad["polys_batch"][0][1] = (bad["polys_batch"][0][1] + 1) % q
    assert not SF.verify(curve_id, m_arr, num_cons, num_vars, X, key_arr, comm_W, comm_E, u, bad)


def test_fast_oracle_at_2_12():
    """The fast oracle alone at a size the Python one cannot reach: complete (its verifier accepts its proof) and sound against a
    tampered statement; a few seconds."""
    import numpy as np

    from oracle import coracle as C
    from oracle import spartan_fast as SF

    curve_id, f, nc, nv, nio = 0, 1, 1 << 12, 1 << 12, 2
    q = R.modulus(f)
    A, B, Cm, W, X = SF.synth_product_instance(f, nc, nv, nio, seed=3)
    key = C.synth_bases(curve_id, nc + 1)
    E = np.zeros((nc, 4), dtype=np.uint64)
    comm_W = SF._aff(curve_id, SF._commit(curve_id, key, W))
    proof = SF.prove(curve_id, (A, B, Cm), nc, nv, X, key, comm_W, None, 1, W, E)
    assert SF.verify(curve_id, (A, B, Cm), nc, nv, X, key, comm_W, None, 1, proof)
    assert not SF.verify(curve_id, (A, B, Cm), nc, nv, X, key, comm_W, None, 2, proof)


def test_batched_oracle_complete_and_sound():
    """oracle/spartan_fast.py prove_batched / verify_batched (the structure of arecibo's BatchedRelaxedR1CSSNARK, which SuperNova compresses
    with: /root/reference/src/proof/supernova.rs:110,293-302): three instances of different shapes and sizes - strict and folded - under
    one key; honest proofs verify, tampered proofs and wrong statements do not."""
    import numpy as np

    from oracle import coracle as C
    from oracle import spartan_fast as SF

    curve_id, curve = 0, "pallas"
    q = R.CURVES[curve]["order"]
    dims = [(16, 64, True, 11), (64, 128, False, 12), (8, 16, True, 13)]  # (num_cons, num_vars, folded, seed): the middle one is the largest
    N = max(max(nc, nv) for nc, nv, _, _ in dims)
    key = C.synth_bases(curve_id, N + 1)
    insts = []
    for nc, nv, folded, seed in dims:
        mats, X, u, W, E = product_instance(curve, nc, nv, 2, seed, folded)
        m_arr, W_arr, E_arr = _to_arrays(mats, X, W, E)
        insts.append(dict(mats=m_arr, num_cons=nc, num_vars=nv, X=X, u=u, W=W_arr, E=E_arr,
                          comm_W=SF._aff(curve_id, SF._commit(curve_id, key, W_arr)), comm_E=SF._aff(curve_id, SF._commit(curve_id, key, E_arr))))
    proof = SF.prove_batched(curve_id, insts, key)
    pub = [{k: v for k, v in it.items() if k not in ("W", "E")} for it in insts]
    assert SF.verify_batched(curve_id, pub, key, proof)
    # one instance alone through the batched protocol as well
    p1 = SF.prove_batched(curve_id, insts[:1], key)
    assert SF.verify_batched(curve_id, pub[:1], key, p1)
    # wrong statements
    for i in range(3):
        bad = copy.deepcopy(pub)
        bad[i]["X"] = [(bad[i]["X"][0] + 1) % q] + bad[i]["X"][1:]
        assert not SF.verify_batched(curve_id, bad, key, proof), i
        bad = copy.deepcopy(pub)
        bad[i]["u"] = (bad[i]["u"] + 1) % q
        assert not SF.verify_batched(curve_id, bad, key, proof), i
    assert not SF.verify_batched(curve_id, [pub[1], pub[0], pub[2]], key, proof)  # the order of the instances is part of the statement
    # tampered proofs
    for field, idx in (("evals_W", 0), ("evals_W", 2), ("evals_E", 1), ("evals_batch", 3)):
        bad = copy.deepcopy(proof)
        bad[field][idx] = (bad[field][idx] + 1) % q
        assert not SF.verify_batched(curve_id, pub, key, bad), (field, idx)
    bad = copy.deepcopy(proof)
    bad["claims_outer"][2][1] = (bad["claims_outer"][2][1] + 1) % q
    assert not SF.verify_batched(curve_id, pub, key, bad)
    bad = copy.deepcopy(proof)
    bad["ipa_a"] = (bad["ipa_a"] + 1) % q
    assert not SF.verify_batched(curve_id, pub, key, bad)
    # an unsatisfied instance in the batch: the honest prover's proof is rejected
    E2 = insts[0]["E"].copy()
    E2[0, 0] += 1
    broken = copy.deepcopy(insts)
    broken[0]["E"] = E2
    broken[0]["comm_E"] = SF._aff(curve_id, SF._commit(curve_id, key, E2))
    p2 = SF.prove_batched(curve_id, broken, key)
    assert not SF.verify_batched(curve_id, [{k: v for k, v in it.items() if k not in ("W", "E")} for it in broken], key, p2)

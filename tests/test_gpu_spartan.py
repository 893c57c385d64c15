"""f3 end to end: the device-assisted Spartan-style prover (lurk_beta_amd/spartan.py over the sum-check, eq, sparse mat-vec, fold,
MSM and inner-product-argument kernels) must produce the oracle prover's proof element for element, and the oracle's VERIFIER must
accept it (and reject it for another statement).  Parity unpinned upstream; not byte-compatible with arecibo (see oracle/spartan_ref.py)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R
from oracle import spartan_ref as S
from tests.test_oracle_spartan import product_instance

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
@pytest.mark.parametrize("num_cons,num_vars,folded", [(8, 16, False), (32, 64, True), (64, 32 * 4, True)])
def test_device_prover_matches_the_oracle_and_verifies(hip, cn, c, num_cons, num_vars, folded):
    from lurk_beta_amd import CommitmentKey
    from lurk_beta_amd.spartan import SpartanProver

    sf, bf = (1, 0) if c == 0 else (0, 1)
    q = R.CURVES[cn]["order"]
    mats, X, u, W, E = product_instance(cn, num_cons, num_vars, 2, 9, folded)
    N = max(num_cons, num_vars)
    B = C.synth_bases(c, N + 1)
    key = [tuple(C.limbs_to_ints(C.from_mont(bf, B[i].reshape(2, 4)))) for i in range(N + 1)]
    ck, ck_c = key[:N], key[N]
    comm_W, comm_E = R.msm_naive(cn, W, ck), R.msm_naive(cn, E, ck)
    want = S.prove(cn, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, W, E)
    to_csr = lambda m: (np.array(m[0], dtype=np.uint64), np.array(m[1], dtype=np.uint64), C.to_mont(sf, C.ints_to_limbs(m[2])))
    prover = SpartanProver(c, q, [to_csr(m) for m in mats], num_cons, num_vars, len(X))
    k = CommitmentKey(c, B[:N])
    d_W, d_E = _dev(C.to_mont(sf, C.ints_to_limbs(W))), _dev(C.to_mont(sf, C.ints_to_limbs(E)))
    cw = k.commit(C.ints_to_limbs(W))
    ce = k.commit(C.ints_to_limbs(E))
    got = prover.prove(X, u, d_W, d_E, _dev(B), cw, ce)
    assert got == want
    # the opening argument under the resident key (never folded) gives the identical proof
    assert prover.prove(X, u, d_W, d_E, _dev(B), cw, ce, key=k) == want
    # ... and so does the whole prover as ONE library call (lurk_hip_spartan_prove_dev), here with a relaxed instance (u != 1, E != 0)
    assert prover.prove(X, u, d_W, d_E, _dev(B), cw, ce, key=k, in_library=True) == want
    assert S.verify(cn, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, got)
    assert not S.verify(cn, mats, num_cons, num_vars, [(X[0] + 1) % q] + X[1:], ck, ck_c, comm_W, comm_E, u, got)
    k.close()
    prover.close()


@pytest.mark.parametrize("cn,c,log_n", [("pallas", 0, 14), ("vesta", 1, 14), ("pallas", 0, 16)])
def test_device_prover_matches_the_fast_oracle_at_2_14_and_2_16(hip, cn, c, log_n):
    """The same end-to-end parity at 2^14 and 2^16 rows and variables: the device-assisted prover's proof equals the proof of the C-backed
    oracle prover (oracle/spartan_fast.py: the protocol of spartan_ref.py, checked identical to it on CPU at small sizes) element for
    element - in all three forms of the opening argument (folded key, resident plain key, resident table key) - and the oracle's
    verifier accepts it and rejects it for another statement.  Reference: /root/reference/src/proof/nova.rs:341-356."""
    from lurk_beta_amd import CommitmentKey
    from lurk_beta_amd.spartan import SpartanProver
    from oracle import spartan_fast as SF

    sf = 1 - c
    q = R.CURVES[cn]["order"]
    nc = nv = 1 << log_n
    A, Bm, Cm, W, X = SF.synth_product_instance(sf, nc, nv, 2, seed=log_n + c)
    E = np.zeros((nc, 4), dtype=np.uint64)
    B = C.synth_bases(c, nc + 1)
    comm_W = SF._aff(c, SF._commit(c, B, W))
    want = SF.prove(c, (A, Bm, Cm), nc, nv, X, B, comm_W, None, 1, W, E)
    assert SF.verify(c, (A, Bm, Cm), nc, nv, X, B, comm_W, None, 1, want)
    mont = lambda M: (M[0], M[1], C.to_mont(sf, M[2]))
    prover = SpartanProver(c, q, [mont(A), mont(Bm), mont(Cm)], nc, nv, len(X))
    d_W, d_E, d_B = _dev(C.to_mont(sf, W)), _dev(E), _dev(B)
    plain = CommitmentKey(c, B[:nc])
    table = CommitmentKey(c, B[:nc], precompute=True)
    cw, ce = plain.commit(W), plain.commit(E)
    got = prover.prove(X, 1, d_W, d_E, d_B, cw, ce)
    assert got == want
    assert prover.prove(X, 1, d_W, d_E, d_B, cw, ce, key=plain) == want
    assert prover.prove(X, 1, d_W, d_E, d_B, cw, ce, key=table) == want
    assert prover.prove(X, 1, d_W, d_E, d_B, cw, ce, key=table, in_library=True) == want   # lurk_hip_spartan_prove_dev: the prover as one library call
    assert prover.prove(X, 1, d_W, d_E, d_B, cw, ce, key=plain, in_library=True) == want
    assert not SF.verify(c, (A, Bm, Cm), nc, nv, [(X[0] + 1) % q] + X[1:], B, comm_W, None, 1, got)
    for k in (plain, table):
        k.close()
    prover.close()


@pytest.mark.parametrize("cn,c,dims", [("pallas", 0, [(16, 64, True, 11), (64, 128, False, 12), (8, 16, True, 13)]),
                                       ("vesta", 1, [(32, 64, True, 21), (8, 32, False, 22)])])
def test_batched_device_prover_matches_the_oracle(hip, cn, c, dims):
    """The batched SNARK (several instances of different shapes under one key, one proof: the structure of arecibo's
    BatchedRelaxedR1CSSNARK, /root/reference/src/proof/supernova.rs:110, 293-302): the device-assisted prover's proof equals the oracle
    prover's (oracle/spartan_fast.py: prove_batched) element for element - opening argument with the folded key and under the resident
    key - and the oracle's verifier accepts it and rejects it for another statement."""
    from lurk_beta_amd import CommitmentKey
    from lurk_beta_amd.spartan import BatchedSpartanProver, SpartanProver
    from oracle import spartan_fast as SF
    from tests.test_oracle_spartan import _to_arrays

    sf = 1 - c
    q = R.CURVES[cn]["order"]
    N = max(max(nc, nv) for nc, nv, _, _ in dims)
    B = C.synth_bases(c, N + 1)
    key = CommitmentKey(c, B[:N])
    insts, provers, dev_insts = [], [], []
    for nc, nv, folded, seed in dims:
        mats, X, u, W, E = product_instance(cn, nc, nv, 2, seed, folded)
        m_arr, W_arr, E_arr = _to_arrays(mats, X, W, E)
        cw, ce = key.commit(W_arr), key.commit(E_arr)
        aff = lambda J: (lambda a: None if a == (0, 0) else a)(C.jac_to_affine(c, J))
        insts.append(dict(mats=m_arr, num_cons=nc, num_vars=nv, X=X, u=u, W=W_arr, E=E_arr, comm_W=aff(cw), comm_E=aff(ce)))
        provers.append(SpartanProver(c, q, [(M[0], M[1], C.to_mont(sf, M[2])) for M in m_arr], nc, nv, len(X)))
        dev_insts.append(dict(X=X, u=u, d_W=_dev(C.to_mont(sf, W_arr)), d_E=_dev(C.to_mont(sf, E_arr)), comm_W=cw, comm_E=ce))
    want = SF.prove_batched(c, insts, B)
    bp = BatchedSpartanProver(provers)
    got = bp.prove(dev_insts, _dev(B))
    assert got == want
    assert bp.prove(dev_insts, _dev(B), key=key) == want
    assert bp.prove(dev_insts, _dev(B), key=key, in_library=True) == want   # lurk_hip_spartan_prove_batch_dev: the batched prover as one library call
    pub = [{k: v for k, v in it.items() if k not in ("W", "E")} for it in insts]
    assert SF.verify_batched(c, pub, B, got)
    pub[0]["X"] = [(pub[0]["X"][0] + 1) % q] + pub[0]["X"][1:]
    assert not SF.verify_batched(c, pub, B, got)
    key.close()
    for p in provers:
        p.close()


def test_batched_device_prover_at_2_12_and_2_14(hip):
    """Two circuits of 2^12 and 2^14 rows (the shape of a SuperNova batch: a small coprocessor circuit beside the step circuit) under one
    resident table key: proof = the oracle's, verifier accepts."""
    from lurk_beta_amd import CommitmentKey
    from lurk_beta_amd.spartan import BatchedSpartanProver, SpartanProver
    from oracle import spartan_fast as SF

    c, sf = 0, 1
    q = R.CURVES["pallas"]["order"]
    sizes = [1 << 12, 1 << 14]
    N = max(sizes)
    B = C.synth_bases(c, N + 1)
    key = CommitmentKey(c, B[:N], precompute=True)
    insts, provers, dev_insts = [], [], []
    for k, n in enumerate(sizes):
        A, Bm, Cm, W, X = SF.synth_product_instance(sf, n, n, 2, seed=30 + k)
        E = np.zeros((n, 4), dtype=np.uint64)
        cw = key.commit(W)
        insts.append(dict(mats=(A, Bm, Cm), num_cons=n, num_vars=n, X=X, u=1, W=W, E=E, comm_W=SF._aff(c, cw), comm_E=None))
        provers.append(SpartanProver(c, q, [(M[0], M[1], C.to_mont(sf, M[2])) for M in (A, Bm, Cm)], n, n, len(X)))
        dev_insts.append(dict(X=X, u=1, d_W=_dev(C.to_mont(sf, W)), d_E=_dev(E), comm_W=cw, comm_E=np.zeros(12, dtype=np.uint64)))
    want = SF.prove_batched(c, insts, B)
    got = BatchedSpartanProver(provers).prove(dev_insts, _dev(B), key=key)
    assert got == want
    assert BatchedSpartanProver(provers).prove(dev_insts, _dev(B), key=key, in_library=True) == want
    assert SF.verify_batched(c, [{k: v for k, v in it.items() if k not in ("W", "E")} for it in insts], B, got)
    key.close()
    for p in provers:
        p.close()

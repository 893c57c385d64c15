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
    assert S.verify(cn, mats, num_cons, num_vars, X, ck, ck_c, comm_W, comm_E, u, got)
    assert not S.verify(cn, mats, num_cons, num_vars, [(X[0] + 1) % q] + X[1:], ck, ck_c, comm_W, comm_E, u, got)
    k.close()
    prover.close()

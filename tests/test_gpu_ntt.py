"""HIP radix-2 NTT against the textbook oracle.  Parity UNPINNED: the reference has no NTT
(SURVEY.md section 0.5).  Needs an MI355X."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("f", [0, 1])
@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 10, 11, 14])
def test_ntt_matches_oracle(hip, f, log_n):
    from lurk_beta_amd import ntt

    n = 1 << log_n
    a = C.synth_scalars(f, 3, 0, n)
    fw = ntt(f, a)
    assert np.array_equal(fw, C.ntt(f, a))
    if log_n <= 6:
        assert C.limbs_to_ints(fw) == R.dft_naive(R.modulus(f), C.limbs_to_ints(a))
    assert np.array_equal(ntt(f, fw, inverse=True), a)


def test_ntt_large_roundtrip_and_linearity(hip):
    from lurk_beta_amd import LurkHipError, ntt

    f, n = 1, 1 << 20
    a = C.synth_scalars(f, 3, 0, n)
    fw = ntt(f, a)
    assert np.array_equal(fw, C.ntt(f, a))
    assert np.array_equal(ntt(f, fw, inverse=True), a)
    with pytest.raises(LurkHipError):
        ntt(2, a[:8])  # BN254 is not offered

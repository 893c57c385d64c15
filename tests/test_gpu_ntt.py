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


def test_full_size_roundtrip_and_linearity_2_24(hip):
    """2^24 elements on the device API: inverse(forward(a)) = a, and forward(a + r b) = forward(a) + r forward(b)
    (checked on the GPU with the fold kernel, then on 4096 sampled positions by the oracle)."""
    import torch

    from lurk_beta_amd import _lib, fold_vec, synth

    lib = _lib.load()
    f, log_n = 1, 24
    n = 1 << log_n
    stream = torch.cuda.current_stream().cuda_stream
    p = R.modulus(f)
    a = synth.scalars(f, 3, 0, n, mont=True)
    b = synth.scalars(f, 4, 0, n, mont=True)
    r = R.uniform_fe(75, 0, p)
    r_mont = C.to_mont(f, C.ints_to_limbs([r]))
    # the NTT takes canonical bytes; Montgomery inputs are canonical bytes of x R, and the transform is linear: fine for both properties
    c = fold_vec(f, a, b, r_mont)
    a0 = a.clone()
    for t in (a, b, c):
        _lib.check(lib.lurk_hip_ntt_dev(f, _lib.ptr(t), log_n, 0, _lib.ptr(stream)))
    lin = fold_vec(f, a, b, r_mont)
    assert torch.equal(lin, c)
    idx = torch.from_numpy(np.random.default_rng(6).integers(0, n, 4096)).cuda()
    ha, hb, hc = (t[idx].cpu().numpy().view(np.uint64) for t in (a, b, c))
    assert np.array_equal(C.from_mont(f, hc), C.axpy(f, C.from_mont(f, ha), C.from_mont(f, hb), r))
    _lib.check(lib.lurk_hip_ntt_dev(f, _lib.ptr(a), log_n, 1, _lib.ptr(stream)))
    assert torch.equal(a, a0)


def test_full_size_2_24_every_output_against_the_oracle(hip):
    """BASELINE-size transform compared DIRECTLY: all 2^24 outputs of the forward NTT equal the oracle's (a consistent error in the
    third pass would survive the round trip and the linearity check above), and the inverse of the oracle's output is the input."""
    import torch

    from lurk_beta_amd import _lib, synth

    lib = _lib.load()
    f, log_n = 1, 24
    n = 1 << log_n
    stream = torch.cuda.current_stream().cuda_stream
    d = synth.scalars(f, 3, 0, n)
    host = C.synth_scalars(f, 3, 0, n)
    assert np.array_equal(d[:4096].cpu().numpy().view(np.uint64), host[:4096])
    _lib.check(lib.lurk_hip_ntt_dev(f, _lib.ptr(d), log_n, 0, _lib.ptr(stream)))
    torch.cuda.synchronize()
    want = C.ntt(f, host)
    assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, 4), want)
    d.copy_(torch.from_numpy(want.view(np.int64)))
    _lib.check(lib.lurk_hip_ntt_dev(f, _lib.ptr(d), log_n, 1, _lib.ptr(stream)))
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, 4), host)

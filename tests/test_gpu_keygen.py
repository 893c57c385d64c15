"""f4: commitment-key generation on the device (SHAKE256 on the host, BLAKE2b / SWU / isogeny per lane) against the oracle's
restatement of arecibo's from_label + pasta_curves' hash_to_curve (oracle/keygen_ref.py; constants pinned mathematically, primitives
by hashlib; no key bytes exist upstream)."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import keygen_ref as K
from oracle import pyref as R

pytestmark = pytest.mark.gpu
CURVES = [("pallas", 0), ("vesta", 1)]


def _pts(c, arr):
    bf = 0 if c == 0 else 1
    out = []
    for row in np.ascontiguousarray(arr).reshape(-1, 8):
        x, y = C.limbs_to_ints(C.from_mont(bf, row.reshape(2, 4)))
        out.append(None if (x, y) == (0, 0) else (x, y))
    return out


@pytest.mark.parametrize("cn,c", CURVES)
def test_from_label_matches_the_oracle(hip, cn, c):
    from lurk_beta_amd.msm import ck_from_label

    n = 300
    got = _pts(c, ck_from_label(c, b"ck", n).cpu().numpy().view(np.uint64))
    assert got == K.from_label(cn, b"ck", n)
    other = _pts(c, ck_from_label(c, b"other label", 5).cpu().numpy().view(np.uint64))
    assert other == K.from_label(cn, b"other label", 5) and other[0] != got[0]
    assert ck_from_label(c, b"ck", 0).shape[0] == 0


@pytest.mark.parametrize("cn,c", CURVES)
def test_hash_to_curve_with_another_domain_and_edge_strings(hip, cn, c):
    import torch

    from lurk_beta_amd import _lib

    msgs = [bytes(32), b"\xff" * 32, bytes(range(32))] + [bytes([i]) * 32 for i in range(1, 30)]
    d_in = torch.frombuffer(bytearray(b"".join(msgs)), dtype=torch.uint8).cuda()
    out = torch.empty((len(msgs), 8), dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().lurk_hip_ck_hash_to_curve_dev(c, b"z.cash:test", _lib.ptr(d_in), len(msgs), _lib.ptr(out), None))
    torch.cuda.synchronize()
    assert _pts(c, out.cpu().numpy().view(np.uint64)) == [K.hash_to_curve(cn, b"z.cash:test", m) for m in msgs]


def test_generated_key_commits(hip):
    """A key generated in place (no host copy) commits to the same point as the oracle's MSM over the oracle's key; 2^16 generated
    points all lie on the curve; the table form agrees."""
    from lurk_beta_amd import CommitmentKey, point_to_affine
    from lurk_beta_amd.msm import ck_from_label

    n = 200
    key = K.from_label("pallas", b"ck", n)
    s = C.limbs_to_ints(C.synth_scalars(1, 180, 1, n))
    want = R.msm_naive("pallas", s, key)
    for pre in (False, True):
        ck = CommitmentKey.from_label(0, b"ck", n, precompute=pre)
        assert point_to_affine(0, ck.commit(C.ints_to_limbs(s))) == want
        ck.close()
    big = ck_from_label(0, b"ck", 1 << 16).cpu().numpy().view(np.uint64)
    assert C.on_curve(0, big) and _pts(0, big[:n]) == key

"""GPU parity of the slot-witness kernels and of the device-side assembly of W (W1) through
the C ABI, against oracle/circuit_ref.py (the restated neptune circuit2 / bellpepper gadgets; pinned by the reference's
slot sizes, the KAT digests and constraint satisfaction - tests/test_oracle_circuit.py)."""
import numpy as np
import pytest

from oracle import circuit_ref as CR
from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu
TYPES = (("commitment", 3), ("hash4", 4), ("hash6", 6), ("hash8", 8))


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


@pytest.mark.parametrize("f", [0, 1, 2])
def test_hash_slot_blocks_match_the_circuit_oracle(hip, f):
    from lurk_beta_amd import witness as W

    p = R.modulus(f)
    for name, ar in TYPES:
        pres = [[R.uniform_fe(70 + ar, i * ar + j, p) for j in range(ar)] for i in range(9)] + [[0] * ar, [p - 1] * ar, [1] * ar]
        PRE = C.ints_to_limbs([x for r in pres for x in r]).reshape(len(pres), ar, 4)
        for mont in (False, True):
            blocks = W.slot_witness(f, ar, C.to_mont(f, PRE.reshape(-1, 4)).reshape(PRE.shape) if mont else PRE, mont=mont)
            got = C.limbs_to_ints(C.from_mont(f, blocks.reshape(-1, 4)))
            size = blocks.shape[1]
            for i, pre in enumerate(pres):
                assert got[i * size:(i + 1) * size] == CR.slot_witness(f, name, pre)[0], (f, name, i, mont)


def test_reference_kat_digest_closes_the_block(hip):
    """(commit 123) = hash3(0, 4, 123), /root/reference/src/lem/tests/eval_tests.rs:1940-1947, and hash8 of zeros, trie/mod.rs:932."""
    from lurk_beta_amd import witness as W

    b = W.slot_witness(2, 3, C.ints_to_limbs([0, 4, 123]).reshape(1, 3, 4))
    assert C.limbs_to_ints(C.from_mont(2, b.reshape(-1, 4)))[-1] == 0x0DF269CC1A453B80D4694FE3E54F0FF2D68BFA6A6DD6320446AF03691112E89D
    b = W.slot_witness(2, 8, np.zeros((1, 8, 4), dtype=np.uint64))
    assert C.limbs_to_ints(C.from_mont(2, b.reshape(-1, 4)))[-1] == 0x1CA5B207085F3F0F324A2E0704B18FFF1CDA2E2D686AA85343FEA91DF77BF35B


@pytest.mark.parametrize("f", [0, 1, 2])
def test_bit_decomp_blocks_match_the_oracle(hip, f):
    from lurk_beta_amd import witness as W

    p = R.modulus(f)
    vals = [0, 1, 2, p - 1, p - 2, p - 3, (1 << 200) + 12345, (1 << 253) + 7, p >> 1] + [R.uniform_fe(75, i, p) for i in range(20)]
    V = C.ints_to_limbs(vals).reshape(len(vals), 1, 4)
    for mont in (False, True):
        blocks = W.slot_witness(f, W.SLOT_BIT_DECOMP, C.to_mont(f, V.reshape(-1, 4)).reshape(V.shape) if mont else V, mont=mont)
        got = C.limbs_to_ints(C.from_mont(f, blocks.reshape(-1, 4)))
        size = blocks.shape[1]
        for i, v in enumerate(vals):
            assert got[i * size:(i + 1) * size] == CR.slot_witness(f, "bit_decomp", [v])[0], (f, i, mont)


def test_large_batch_takes_the_lane_per_hash_kernel(hip):
    """More slots than the lane-cooperative kernel handles (rc = 900: 12 600 hash4 slots per step): sampled blocks vs the oracle,
    every digest vs the batch hasher."""
    from lurk_beta_amd import poseidon_batch, witness as W

    f, ar, n = 1, 4, 12600
    PRE = C.synth_scalars(f, 77, 1, n * ar).reshape(n, ar, 4)
    blocks = W.slot_witness(f, ar, PRE)
    digests = C.from_mont(f, np.ascontiguousarray(blocks[:, -1, :]))
    assert np.array_equal(digests, poseidon_batch(f, ar, PRE))
    assert np.array_equal(C.from_mont(f, np.ascontiguousarray(blocks[:, :ar, :]).reshape(-1, 4)), PRE.reshape(-1, 4))
    for i in (0, 1, 63, 64, 4097, n - 1):
        got = C.limbs_to_ints(C.from_mont(f, np.ascontiguousarray(blocks[i])))
        assert got == CR.slot_witness(f, "hash4", C.limbs_to_ints(PRE[i]))[0], i


def test_multiframe_w_is_assembled_on_the_device_and_committed(hip):
    """W1: W = [globals | frame 0 | ... ], frame = [14 hash4, 6 hash8, 1 commitment, 3 bit-decomp blocks | body]
    (multiframe.rs:699-702, circuit.rs:1429-1451, eval.rs:1960-1964) assembled in HBM == the oracle's W element by element;
    the device-resident W then goes straight into the Pedersen commitment (no PCIe) == the oracle's MSM over the oracle's W."""
    import torch

    from lurk_beta_amd import CommitmentKey, MultiFrameWitness, point_to_affine, synth

    f, frames, glob, body = 1, 5, 37, 1311
    p = R.modulus(f)
    mf = MultiFrameWitness(f, frames, glob, body)
    assert mf.slots_len == CR.slot_witness_size(f, "hash4") * 14 + CR.slot_witness_size(f, "hash8") * 6 + CR.slot_witness_size(f, "commitment") + 3 * 298
    rng = np.random.default_rng(5)
    pre = {}
    for name, st in (("hash4", 4), ("hash8", 8), ("commitment", 3), ("bit_decomp", 1)):
        cnt = mf.counts[name]
        vals = C.synth_scalars(f, 80 + st, 1, frames * cnt * st).reshape(frames * cnt, st, 4)
        if cnt > 2:  # dummy slots: all-zero preimages (allocate_slot's None branch, circuit.rs:296-312)
            vals[rng.integers(0, frames * cnt, frames)] = 0
        pre[name] = vals
    g_host = C.to_mont(f, C.synth_scalars(f, 90, 0, glob))
    b_host = C.to_mont(f, C.synth_scalars(f, 91, 1, frames * body)).reshape(frames, body, 4)
    d_w = torch.zeros((mf.w_len, 4), dtype=torch.int64, device="cuda")
    d_pre = {k: _dev(C.to_mont(f, v.reshape(-1, 4))) for k, v in pre.items()}
    mf.assemble(d_w, d_pre, g_host, b_host, mont=True)
    d_w_b = torch.zeros_like(d_w)
    mf.assemble(d_w_b, d_pre, g_host, b_host, mont=True, per_type_offsets=True)  # explicit per-slot offsets: same vector
    torch.cuda.synchronize()
    assert torch.equal(d_w, d_w_b)
    got = C.from_mont(f, d_w.cpu().numpy().view(np.uint64))
    want = C.limbs_to_ints(C.from_mont(f, g_host))
    for fr in range(frames):
        rows = {k: [C.limbs_to_ints(x) for x in v[fr * mf.counts[k]:(fr + 1) * mf.counts[k]]] for k, v in pre.items()}
        rows["hash6"] = []
        want += CR.frame_slot_block(f, rows) + C.limbs_to_ints(C.from_mont(f, b_host[fr]))
    assert len(want) == mf.w_len
    assert C.limbs_to_ints(got) == want
    # commit(W) on the resident vector
    d_bases = synth.bases(0, mf.w_len)
    ck = CommitmentKey(0, d_bases, n=mf.w_len, device=True)
    com = point_to_affine(0, ck.commit_device(d_w, mf.w_len, is_mont=True))
    assert com == C.jac_to_affine(0, C.msm_pippenger(0, C.synth_bases(0, mf.w_len), C.ints_to_limbs(want)))
    ck.close()


def test_bad_arguments(hip):
    from lurk_beta_amd import LurkHipError, witness as W

    with pytest.raises(LurkHipError):
        W.slot_witness(1, 5, np.zeros((1, 5, 4), dtype=np.uint64))
    assert W.slot_witness(1, 4, np.zeros((0, 4, 4), dtype=np.uint64)).shape[0] == 0

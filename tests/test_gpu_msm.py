"""Parity of the HIP Pedersen MSM (through the C ABI) against the oracle.  Needs an MI355X.

The reference holds no golden commitment (MSM parity is unpinned), so the
anchors are: naive double-and-add (oracle/pyref.py, oracle/oracle.c), the oracle's Pippenger, and -
at BASELINE.json's full sizes - the discrete-log checksum  sum s_i [k_i]G == [sum s_i k_i] G."""
import numpy as np
import pytest

from oracle import coracle as C
from oracle import pyref as R

pytestmark = pytest.mark.gpu

CURVES = [("pallas", 0), ("vesta", 1)]


def _sf(c):  # scalar field id of curve c
    return 1 if c == 0 else 0


def test_synth_generators_match_oracle(hip):
    import torch

    from lurk_beta_amd import synth

    for f in (0, 1, 2):
        for dist in (0, 1):
            got = synth.scalars(f, 5, dist, 3000, first=17).cpu().numpy().view(np.uint64)
            assert np.array_equal(got, C.synth_scalars(f, 5, dist, 3000, first=17)), (f, dist)
        gm = synth.scalars(f, 5, 0, 100, mont=True).cpu().numpy().view(np.uint64)
        assert np.array_equal(gm, C.to_mont(f, C.synth_scalars(f, 5, 0, 100)))
    for c in (0, 1):
        got = synth.bases(c, 700, first=3).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, C.synth_bases(c, 700, first=3)), c
        assert C.on_curve(c, got[:50])


@pytest.mark.parametrize("cn,c", CURVES)
@pytest.mark.parametrize("n", [1, 2, 7, 100, 1000])
def test_small_msm_matches_naive(hip, cn, c, n):
    from lurk_beta_amd import msm, point_to_affine

    B = C.synth_bases(c, n)
    S = C.synth_scalars(_sf(c), 1, 0, n)
    want = C.jac_to_affine(c, C.msm_naive(c, B, S))
    assert point_to_affine(c, msm(c, B, S, is_mont=False)) == want
    assert point_to_affine(c, msm(c, B, C.to_mont(_sf(c), S), is_mont=True)) == want


def test_pyref_cross_check(hip):
    """Tiny case against the pure-Python group law (independent of the C oracle)."""
    from lurk_beta_amd import msm, point_to_affine

    pts = R.synth_bases("pallas", 5)
    s = [R.uniform_fe(1, i, R.PALLAS_Q) for i in range(5)]
    B = C.synth_bases(0, 5)
    assert point_to_affine(0, msm(0, B, C.ints_to_limbs(s))) == R.msm_naive("pallas", s, pts)


@pytest.mark.parametrize("cn,c", CURVES)
def test_edge_cases(hip, cn, c):
    from lurk_beta_amd import msm, point_to_affine

    q = R.CURVES[cn]["order"]
    n = 64
    B = C.synth_bases(c, n)
    B[1] = 0                      # identity base (0,0)
    B[3] = B[2]                   # repeated base -> doubling inside a bucket when scalars agree
    B[5] = B[4]
    s = [R.uniform_fe(9, i, q) for i in range(n)]
    s[2] = s[3] = 12345           # same bucket, same point: exercises the doubling branch
    s[4], s[5] = 777, q - 777     # P and -P in the same bucket: identity mid-chain
    s[6] = 0
    s[7] = 1
    s[8] = q - 1
    s[9] = 0x8000                 # digit exactly 2^15 (largest positive bucket)
    s[10] = 0x8001                # first value that recodes to a negative digit with carry
    s[11] = (1 << 254) | 0xFFFF   # carries rippling from the bottom, top window in use
    s[12] = int("ffff" * 15, 16) % q
    S = C.ints_to_limbs(s)
    want = C.jac_to_affine(c, C.msm_naive(c, B, S))
    assert point_to_affine(c, msm(c, B, S)) == want
    # all-zero scalars -> identity, encoded z = 0 / affine (0,0)
    out = msm(c, B, np.zeros((n, 4), dtype=np.uint64))
    assert point_to_affine(c, out) == (0, 0) and not out[8:].any()
    # empty input
    assert point_to_affine(c, msm(c, np.zeros((0, 8), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))) == (0, 0)
    # every scalar identical: one hot bucket per window (the workgroup tree path)
    n2 = 5000
    B2 = C.synth_bases(c, n2)
    S2 = np.tile(C.ints_to_limbs([s[0]]), (n2, 1))
    assert point_to_affine(c, msm(c, B2, S2)) == C.jac_to_affine(c, C.msm_pippenger(c, B2, S2))
    # every base identical and every scalar 1: n * P through the doubling path
    B3 = np.tile(B[:1], (300, 1))
    S3 = C.ints_to_limbs([1] * 300)
    assert point_to_affine(c, msm(c, B3, S3)) == C.jac_to_affine(c, C.msm_naive(c, B3, S3))


@pytest.mark.parametrize("cn,c", CURVES)
@pytest.mark.parametrize("dist", [0, 1])
@pytest.mark.parametrize("log_n", [14, 16])
def test_medium_msm_matches_oracle_pippenger(hip, cn, c, dist, log_n):
    from lurk_beta_amd import msm, point_to_affine

    n = 1 << log_n
    B = C.synth_bases(c, n)
    S = C.synth_scalars(_sf(c), 1, dist, n)
    assert point_to_affine(c, msm(c, B, S)) == C.jac_to_affine(c, C.msm_pippenger(c, B, S))


@pytest.mark.parametrize("cn,c", CURVES)
def test_commitment_key_prefix_and_precompute(hip, cn, c):
    """CE::commit(ck, v) uses ck[..v.len()]; the precomputed-table context must agree bit for bit."""
    from lurk_beta_amd import CommitmentKey, point_to_affine

    n = 4096
    sf = _sf(c)
    B = C.synth_bases(c, n)
    ck = CommitmentKey(c, B)
    ckp = CommitmentKey(c, B, precompute=True)
    ckw = {w: CommitmentKey(c, B, precompute=True, window_bits=w) for w in (16, 17, 18, 19, 20)}  # (the reduction takes a different mix of pair, single and butterfly launches at every width)
    for m, dist in ((n, 0), (n, 1), (1000, 0), (1, 0), (0, 0)):
        S = C.synth_scalars(sf, 3, dist, m)
        want = C.jac_to_affine(c, C.msm_pippenger(c, B[:m], S)) if m else (0, 0)
        assert point_to_affine(c, ck.commit(S)) == want, (m, dist)
        assert point_to_affine(c, ckp.commit(S)) == want, ("precompute", m, dist)
        for w, k in ckw.items():
            assert point_to_affine(c, k.commit(S)) == want, ("precompute", w, m, dist)
        assert point_to_affine(c, ck.commit(C.to_mont(sf, S), is_mont=True)) == want
    # scalars that stress the signed-digit recoding at every window width
    q = R.CURVES[cn]["order"]
    edge = [0, 1, q - 1, (1 << 254) | 0xFFFFF, int("f" * 63, 16) % q, 1 << 19, (1 << 19) + 1, (1 << 17) + 1, 0x80000, 0x7FFFF]
    S = C.ints_to_limbs(edge)
    want = C.jac_to_affine(c, C.msm_naive(c, B[: len(edge)], S))
    for k in [ck, ckp] + list(ckw.values()):
        assert point_to_affine(c, k.commit(S)) == want
    for k in ckw.values():
        k.close()
    from lurk_beta_amd import LurkHipError

    with pytest.raises(LurkHipError):
        ck.commit(C.synth_scalars(sf, 3, 0, n + 1))
    ck.close()
    ckp.close()


@pytest.mark.parametrize("cn,c", CURVES)
def test_async_slots(hip, cn, c):
    """submit/wait: three commitments in flight on one context give the same bytes as the synchronous call."""
    import torch

    from lurk_beta_amd import CommitmentKey, LurkHipError, point_to_affine, synth

    n = 1 << 14
    d_bases = synth.bases(c, n)
    scal = [synth.scalars(_sf(c), 10 + j, j % 2, n, mont=True) for j in range(5)]
    torch.cuda.synchronize()
    want0 = C.jac_to_affine(c, C.msm_pippenger(c, C.synth_bases(c, n), C.synth_scalars(_sf(c), 10, 0, n)))
    for pre in (False, True):
        ck = CommitmentKey(c, d_bases, n=n, device=True, precompute=pre)
        want = [ck.commit_device(sc, n, is_mont=True) for sc in scal]
        got = [None] * 5
        for j in range(5):
            slot = j % 3
            if j >= 3:
                got[j - 3] = ck.wait(slot)
            ck.submit_device(slot, scal[j], n, is_mont=True, stream=torch.cuda.current_stream().cuda_stream)
        for j in range(2, 5):
            got[j] = ck.wait(j % 3)
        for a, b in zip(got, want):
            assert point_to_affine(c, a) == point_to_affine(c, b)
        assert point_to_affine(c, got[0]) == want0
        with pytest.raises(LurkHipError):
            ck.wait(0)  # nothing pending
        ck.submit_device(1, scal[0], n, is_mont=True)
        with pytest.raises(LurkHipError):
            ck.submit_device(1, scal[1], n, is_mont=True)  # slot busy
        ck.wait(1)
        ck.close()


@pytest.mark.parametrize("log_n", [14, 20])
def test_submit_scheduling_classes_and_four_slots(hip, log_n):
    """lurk_hip_msm_ctx_submit_dev_mode: the class changes how the accumulation is launched (plain / persistent with one or
    two waves per SIMD, raised wave priority, start behind the foreground sort), never the result; all four slots in flight."""
    import torch

    from lurk_beta_amd import CommitmentKey, LurkHipError, point_to_affine, synth

    c, n = 0, 1 << log_n
    d_bases = synth.bases(c, n)
    scal = [synth.scalars(_sf(c), 40 + j, j % 2, n, mont=True) for j in range(4)]
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    for pre in (False, True):
        ck = CommitmentKey(c, d_bases, n=n, device=True, precompute=pre)
        ck.reserve(n, 4)
        want = [point_to_affine(c, ck.commit_device(sc, n, is_mont=True)) for sc in scal]
        # 1 = foreground, 2 = background, 3 = follow (round 6: behind the latest foreground commitment's accumulation; with nothing to follow it just runs)
        for modes in ((1, 2, 2, 0), (2, 1, 0, 1), (2, 2, 2, 2), (1, 1, 1, 1), (1, 3, 3, 0), (3, 1, 3, 1), (3, 3, 3, 3)):
            for slot in range(4):
                ck.submit_device(slot, scal[slot], n, is_mont=True, stream=s, mode=modes[slot])
            assert [point_to_affine(c, ck.wait(slot)) for slot in range(4)] == want, modes
        with pytest.raises(LurkHipError):
            ck.submit_device(0, scal[0], n, is_mont=True, stream=s, mode=4)  # unknown class
        with pytest.raises(LurkHipError):
            ck.submit_device(6, scal[0], n, is_mont=True, stream=s)  # LURK_MSM_SLOTS = 6 slots
        ck.close()


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_pasta_msm_symbol_names(hip, cn, c):
    """mult_pippenger_{pallas,vesta}: the names and the signature pasta-msm's Rust side binds (void return, bool is_mont)."""
    import ctypes

    from lurk_beta_amd import _lib, point_to_affine

    n = 3000
    B = C.synth_bases(c, n)
    S = C.synth_scalars(_sf(c), 77, 0, n)
    want = C.jac_to_affine(c, C.msm_pippenger(c, B, S))
    fn = getattr(_lib.load(), f"mult_pippenger_{cn}")
    for is_mont, scal in ((False, S), (True, C.to_mont(_sf(c), S))):
        out = np.zeros(12, dtype=np.uint64)
        fn(_lib.ptr(out), _lib.ptr(B), n, _lib.ptr(np.ascontiguousarray(scal)), ctypes.c_bool(is_mont))
        assert point_to_affine(c, out) == want


@pytest.mark.parametrize("cn,c", [("pallas", 0), ("vesta", 1)])
def test_pasta_msm_gpu_symbol_names(hip, cn, c):
    """cuda_pippenger_{pallas,vesta}: pasta-msm's GPU symbols (sppark convention): RustError {code, message} BY VALUE;
    code 0 / NULL message on success, the library's code and a malloc'd message on failure (freed here as the Rust side would)."""
    import ctypes

    from lurk_beta_amd import _lib, point_to_affine

    n = 3000
    B = C.synth_bases(c, n)
    S = C.synth_scalars(_sf(c), 78, 1, n)
    want = C.jac_to_affine(c, C.msm_pippenger(c, B, S))
    fn = getattr(_lib.load(), f"cuda_pippenger_{cn}")
    for is_mont, scal in ((False, S), (True, C.to_mont(_sf(c), S))):
        out = np.zeros(12, dtype=np.uint64)
        err = fn(_lib.ptr(out), _lib.ptr(B), n, _lib.ptr(np.ascontiguousarray(scal)), ctypes.c_bool(is_mont))
        assert err.code == 0 and not err.message
        assert point_to_affine(c, out) == want
    err = fn(None, _lib.ptr(B), n, _lib.ptr(S), ctypes.c_bool(False))  # null output: an error value, not an abort
    assert err.code != 0 and err.message
    assert b"null" in ctypes.string_at(err.message)
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(err.message)


def test_oneshot_key_cache(hip):
    """lurk_hip_msm_oneshot_key_cache: same pointer + same sampled points -> the device copy is reused (also for a prefix);
    new contents at the same address -> detected, uploaded again."""
    from lurk_beta_amd import _lib, msm, point_to_affine

    lib = _lib.load()
    c, n = 0, 20000
    B = np.ascontiguousarray(C.synth_bases(c, n))
    S1, S2 = C.synth_scalars(1, 90, 0, n), C.synth_scalars(1, 91, 1, n)
    want = lambda bases, sc: C.jac_to_affine(c, C.msm_pippenger(c, bases, sc))
    _lib.check(lib.lurk_hip_msm_oneshot_key_cache(1))
    try:
        assert point_to_affine(c, msm(c, B, S1)) == want(B, S1)           # uploads and remembers
        assert point_to_affine(c, msm(c, B, S2)) == want(B, S2)           # hit
        assert point_to_affine(c, msm(c, B[:7000], S1[:7000])) == want(B[:7000], S1[:7000])  # prefix of the cached key: hit
        B2 = C.synth_bases(c, 2 * n)[n:]
        B[:] = B2                                                         # same address, other key
        assert point_to_affine(c, msm(c, B, S1)) == want(B2, S1)          # miss detected by the samples
        assert point_to_affine(c, msm(c, B, S2)) == want(B2, S2)
        B3 = np.ascontiguousarray(C.synth_bases(c, n + 1000))             # another buffer, longer
        S3 = C.synth_scalars(1, 92, 0, n + 1000)
        assert point_to_affine(c, msm(c, B3, S3)) == want(B3, S3)
    finally:
        _lib.check(lib.lurk_hip_msm_oneshot_key_cache(0))
    assert point_to_affine(c, msm(c, B, S1)) == want(B, S1)               # cache off again: plain path


def test_foreground_commitments_after_buffer_reuse(hip):
    """Foreground-class commitments (submit mode 1) on one slot: new scalars written into the SAME device buffer give new results,
    another buffer / a shorter length likewise, under a plain and under a table key (nothing of an earlier submission is replayed)."""
    import torch

    from lurk_beta_amd import CommitmentKey, point_to_affine, synth

    c, n = 0, 1 << 14
    d_bases = synth.bases(c, n)
    bufs = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(2)]
    s = torch.cuda.current_stream().cuda_stream
    for pre in (False, True):
        ck = CommitmentKey(c, d_bases, n=n, device=True, precompute=pre, window_bits=16 if pre else 0)  # the bucket pipeline, not the small form
        seen = set()
        for j in range(6):
            sc = synth.scalars(1, 60 + j, j % 2, n, mont=True)
            m = n if j < 4 else n // 2
            buf = bufs[j % 2]
            buf.copy_(sc)
            torch.cuda.synchronize()
            want = C.jac_to_affine(c, C.msm_pippenger(c, C.synth_bases(c, m), C.synth_scalars(1, 60 + j, j % 2, m)))
            ck.submit_device(1, buf, m, is_mont=True, stream=s, mode=1)
            got = point_to_affine(c, ck.wait(1))
            assert got == want, (pre, j)
            seen.add(got)
        assert len(seen) == 6
        ck.close()


def test_point_sum(hip):
    from lurk_beta_amd import msm, point_sum, point_to_affine

    n = 512
    B = C.synth_bases(0, n)
    S = C.synth_scalars(1, 4, 0, n)
    parts = np.stack([msm(0, B[i::4], S[i::4]) for i in range(4)])
    assert point_to_affine(0, point_sum(0, parts)) == C.jac_to_affine(0, C.msm_pippenger(0, B, S))


@pytest.mark.parametrize("c,log_n,dist,precompute", [(0, 20, 0, False), (0, 20, 1, False), (0, 20, 1, True), (0, 22, 0, False), (0, 22, 1, True),
                                                     (0, 22, 0, True), (0, 23, 0, True), (0, 23, 1, False),
                                                     (1, 20, 0, False), (1, 20, 1, True), (1, 22, 0, True), (1, 22, 1, False)])
def test_full_size_dlog_checksum(hip, c, log_n, dist, precompute):
    """BASELINE.json sizes (2^20, 2^22) and one size beyond (2^23: the rc = 900 step circuit, where the partitions
    of sort pass 2 no longer fit their LDS stage and take the direct-scatter path): inputs generated in HBM, result
    checked bit-exactly by the size-independent identity  sum_i s_i [k_i]G = [sum_i s_i k_i mod q] G."""
    import torch

    from lurk_beta_amd import CommitmentKey, point_to_affine, synth

    n = 1 << log_n
    sf = _sf(c)
    d_bases = synth.bases(c, n)
    d_scalars = synth.scalars(sf, 1, dist, n, mont=True)
    torch.cuda.synchronize()
    ck = CommitmentKey(c, d_bases, n=n, device=True, precompute=precompute)
    got = point_to_affine(c, ck.commit_device(d_scalars, n, is_mont=True))
    k = C.synth_base_scalars(c, n)
    s = C.synth_scalars(sf, 1, dist, n)
    want = C.jac_to_affine(c, C.gen_mul(c, C.dot(sf, k, s)))
    assert got == want
    ck.close()


@pytest.mark.parametrize("cn,c", CURVES)
def test_oneshot_pasta_msm_entry_point_at_2_20(hip, cn, c):
    """The literal drop-in symbols lurk_hip_msm_{pallas,vesta}(out, points, npoints, scalars, is_mont) - what an
    unmodified arecibo binds instead of pasta-msm's mult_pippenger_* - with HOST buffers at BASELINE configs[1]'s
    size, checked by the discrete-log checksum."""
    from lurk_beta_amd import msm, point_to_affine, synth

    n = 1 << 20
    sf = _sf(c)
    B = synth.bases(c, n).cpu().numpy().view(np.uint64)
    S = C.to_mont(sf, C.synth_scalars(sf, 1, 1, n))
    got = point_to_affine(c, msm(c, B, S, is_mont=True))
    want = C.jac_to_affine(c, C.gen_mul(c, C.dot(sf, C.synth_base_scalars(c, n), C.synth_scalars(sf, 1, 1, n))))
    assert got == want


@pytest.mark.parametrize("cn,c", CURVES)
def test_multi_device_key_one_process(hip, cn, c):
    """lurk_hip_msm_multi_*: one process, a device list (the single GPU of this box listed 2 and 3 times), slices of
    the key on every list entry, partials summed on the host == the oracle; host-pointer and device-pointer commits,
    prefix commits ending inside each slice, plain and table keys."""
    import torch

    from lurk_beta_amd import LurkHipError, MultiCommitmentKey, point_to_affine

    n = 10007
    sf = _sf(c)
    B = C.synth_bases(c, n)
    S = C.synth_scalars(sf, 41, 1, n)
    for devices, pre in (([0, 0], False), ([0, 0, 0], True), ([0], False)):
        mk = MultiCommitmentKey(c, B, devices, precompute=pre)
        sh = mk.shards()
        assert [d for d, _, _ in sh] == devices and sh[0][1] == 0 and sum(cnt for _, _, cnt in sh) == n
        assert all(sh[i][1] + sh[i][2] == sh[i + 1][1] for i in range(len(sh) - 1))
        for m in (n, 17, sh[0][2], sh[0][2] + 1, n - 1, 0):
            want = C.jac_to_affine(c, C.msm_pippenger(c, B[:m], S[:m])) if m else (0, 0)
            assert point_to_affine(c, mk.commit(S[:m])) == want, (devices, pre, m)
            assert point_to_affine(c, mk.commit(C.to_mont(sf, S[:m]), is_mont=True)) == want
        d_slices = [torch.from_numpy(np.ascontiguousarray(C.to_mont(sf, S[f:f + cnt])).view(np.int64)).cuda() for _, f, cnt in sh]
        torch.cuda.synchronize()
        assert point_to_affine(c, mk.commit_device(d_slices, n, is_mont=True)) == C.jac_to_affine(c, C.msm_pippenger(c, B, S))
        with pytest.raises(LurkHipError):
            mk.commit(C.synth_scalars(sf, 41, 0, n + 1))
        mk.close()
    with pytest.raises(LurkHipError):
        MultiCommitmentKey(c, B, [0, 4096])
    # LURK_MSM_FLAG_AUTO_SLICES: only as many of the listed devices as leave every slice the threshold's points (2^20 by default: one
    # slice here; 2^12 from the environment: the first two of the three), same commitment
    import os

    want = C.jac_to_affine(c, C.msm_pippenger(c, B, S))
    mk = MultiCommitmentKey(c, B, [0, 0, 0], auto_slices=True)
    assert len(mk.shards()) == 1 and point_to_affine(c, mk.commit(S)) == want
    mk.close()
    os.environ["LURK_MSM_MULTI_MIN_SLICE_LOG"] = "12"
    try:
        mk = MultiCommitmentKey(c, B, [0, 0, 0], auto_slices=True)
        assert len(mk.shards()) == 2 and sum(cnt for _, _, cnt in mk.shards()) == n and point_to_affine(c, mk.commit(S)) == want
        mk.close()
    finally:
        del os.environ["LURK_MSM_MULTI_MIN_SLICE_LOG"]


def test_multi_device_key_commitments_in_flight(hip):
    """lurk_hip_msm_multi_submit_dev / _wait: two commitments in flight on every slice of a device list (the folding step's W2 and T),
    each slice's scalars produced on a stream of its own, == the oracle; prefixes that end inside a slice; a busy slot is refused and
    leaves the key usable."""
    import torch

    from lurk_beta_amd import LurkHipError, MultiCommitmentKey, point_to_affine

    c, sf, n = 0, 1, 70001
    B = C.synth_bases(c, n)
    S = [C.synth_scalars(sf, 51 + k, k % 2, n) for k in range(3)]
    for devices, pre in (([0, 0], True), ([0, 0, 0, 0], False)):
        mk = MultiCommitmentKey(c, B, devices, precompute=pre, window_bits=16 if pre else 0)
        sh = mk.shards()
        streams = [torch.cuda.Stream() for _ in sh]
        for m_w, m_t in ((n, n - 5), (sh[0][2] + 3, 11), (n, 0)):
            slices = []
            for k, m in ((0, m_w), (1, m_t)):
                row = []
                for (_, f, cnt), st in zip(sh, streams):
                    with torch.cuda.stream(st):  # produced on the slice's own stream: the submit must order itself behind it
                        row.append(torch.from_numpy(np.ascontiguousarray(C.to_mont(sf, S[k][f:f + cnt])).view(np.int64)).cuda(non_blocking=True) + 0)
                slices.append(row)
            raw = [st.cuda_stream for st in streams]
            mk.submit_device(0, slices[0], m_w, is_mont=True, streams=raw, mode=1)
            mk.submit_device(1, slices[1], m_t, is_mont=True, streams=raw, mode=1)
            with pytest.raises(LurkHipError, match="busy"):
                mk.submit_device(1, slices[0], m_w, is_mont=True, streams=raw)
            got_w, got_t = mk.wait(0), mk.wait(1)
            assert point_to_affine(c, got_w) == C.jac_to_affine(c, C.msm_pippenger(c, B[:m_w], S[0][:m_w])), (devices, m_w)
            assert point_to_affine(c, got_t) == (C.jac_to_affine(c, C.msm_pippenger(c, B[:m_t], S[1][:m_t])) if m_t else (0, 0)), (devices, m_t)
            with pytest.raises(LurkHipError):
                mk.wait(0)
        assert point_to_affine(c, mk.commit(S[2])) == C.jac_to_affine(c, C.msm_pippenger(c, B, S[2]))  # the synchronous form still works
        mk.close()


def test_multi_device_full_size(hip):
    """2^22 points over the device list [0, 0] through the host-pointer commit, dlog checksum."""
    from lurk_beta_amd import MultiCommitmentKey, point_to_affine, synth

    n = 1 << 22
    B = synth.bases(0, n).cpu().numpy().view(np.uint64)
    S = C.synth_scalars(1, 1, 0, n)
    mk = MultiCommitmentKey(0, B, [0, 0], precompute=True)
    got = point_to_affine(0, mk.commit(C.to_mont(1, S), is_mont=True))
    assert got == C.jac_to_affine(0, C.gen_mul(0, C.dot(1, C.synth_base_scalars(0, n), S)))
    mk.close()


def test_thread_safety(hip):
    """Entry points are called from rayon worker threads in the reference: concurrent
    commits on one context, on two contexts, and concurrent Poseidon batches must all be exact."""
    import threading

    from lurk_beta_amd import CommitmentKey, point_to_affine, poseidon_batch

    n = 1 << 13
    B = C.synth_bases(0, n)
    cks = [CommitmentKey(0, B), CommitmentKey(0, B, precompute=True)]
    scal = [C.synth_scalars(1, 20 + j, j % 2, n) for j in range(6)]
    want = [C.jac_to_affine(0, C.msm_pippenger(0, B, s)) for s in scal]
    pre = C.synth_scalars(1, 30, 0, 8 * 4000).reshape(4000, 8, 4)
    want_h = C.poseidon_batch(1, 8, pre)
    errors = []

    def work(j):
        try:
            for _ in range(3):
                got = point_to_affine(0, cks[j % 2].commit(scal[j]))
                if got != want[j]:
                    errors.append(("msm", j))
                if not np.array_equal(poseidon_batch(1, 8, pre), want_h):
                    errors.append(("poseidon", j))
        except Exception as e:  # noqa: BLE001
            errors.append((j, repr(e)))

    ts = [threading.Thread(target=work, args=(j,)) for j in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for ck in cks:
        ck.close()


@pytest.mark.parametrize("cn,c", CURVES)
def test_key_files_round_trip(hip, tmp_path, cn, c):
    """f4: a resident key saved and mapped back (bases only / with its table; loaded plain / precomputed from either) commits to
    the same bytes as the oracle; foreign and truncated files are refused."""
    from lurk_beta_amd import CommitmentKey, LurkHipError, point_to_affine

    n = 5000
    sf = _sf(c)
    B = C.synth_bases(c, n)
    B[7] = 0  # an identity base survives the batched-inversion table build
    S = C.synth_scalars(sf, 55, 1, n)
    want = C.jac_to_affine(c, C.msm_pippenger(c, B, S))
    ck = CommitmentKey(c, B, precompute=True, window_bits=16)  # the window table of the bucket pipeline (a key this small would otherwise
    assert point_to_affine(c, ck.commit(S)) == want             # take the small-commitment form, whose table is never written: test_gpu_msm_small.py)
    bases_only, with_table = str(tmp_path / "ck.bin"), str(tmp_path / "ck_table.bin")
    ck.save(bases_only)
    ck.save(with_table, with_table=True)
    import os

    windows = -(-256 // ck.info()["window_bits"])
    assert ck.info()["window_bits"] == 16
    assert os.path.getsize(bases_only) == 64 + 64 * n and os.path.getsize(with_table) == 64 + windows * 64 * n
    for path in (bases_only, with_table):
        for pre in (False, True):
            k2 = CommitmentKey.load(path, precompute=pre)
            info = k2.info()
            assert (info["curve"], info["npoints"], info["precomputed"]) == (c, n, pre)
            # the file's own table is adopted as it is (16-bit windows); bases alone, loaded with the flag, take the small form (8-bit)
            assert info["window_bits"] == (16 if not pre or path == with_table else 8)
            assert point_to_affine(c, k2.commit(S)) == want, (path, pre)
            assert point_to_affine(c, k2.commit(S[:777])) == C.jac_to_affine(c, C.msm_pippenger(c, B[:777], S[:777]))
            k2.close()
    ck.close()
    bad = str(tmp_path / "bad.bin")
    with open(with_table, "rb") as fh, open(bad, "wb") as out:
        out.write(fh.read()[:-64])  # truncated
    with pytest.raises(LurkHipError):
        CommitmentKey.load(bad)
    with open(bad, "wb") as out:
        out.write(b"not a key file" * 10)
    with pytest.raises(LurkHipError):
        CommitmentKey.load(bad)
    with pytest.raises(LurkHipError):
        CommitmentKey.load(str(tmp_path / "missing.bin"))


@pytest.mark.parametrize("cn,c", CURVES)
def test_pair_of_commitments_in_one_pass(hip, cn, c):
    """lurk_hip_msm_ctx_submit_pair_dev / wait_pair: ONE scalar vector, bit sel of the index splits it into two commitments (the L and R of
    an inner-product-argument round have disjoint supports) - two key spaces of one sort / accumulate / reduce.  Must equal the two
    commitments made separately, for every selector bit, with zero halves, and with two pairs in flight."""
    import torch

    from lurk_beta_amd import CommitmentKey, LurkHipError, point_to_affine

    sf, n = _sf(c), 1 << 17
    B = C.synth_bases(c, n)
    key = CommitmentKey(c, B, precompute=True)  # 2^17 points: the window-table form
    assert key.supports_pairs()
    S = C.synth_scalars(sf, 61, 0, n)
    d_s = torch.from_numpy(C.to_mont(sf, S).view(np.int64)).cuda()
    idx = np.arange(n)
    for bit in (0, 1, 7, 16):
        hi_mask = ((idx >> bit) & 1).astype(bool)
        lo_v, hi_v = S.copy(), S.copy()
        lo_v[hi_mask] = 0
        hi_v[~hi_mask] = 0
        key.submit_pair_device(0, d_s, n, bit, is_mont=True)
        lo, hi = key.wait_pair(0)
        assert point_to_affine(c, lo) == C.jac_to_affine(c, C.msm_fast(c, B, lo_v)), bit
        assert point_to_affine(c, hi) == C.jac_to_affine(c, C.msm_fast(c, B, hi_v)), bit
    # one side all zero; a prefix of the key; two pairs in flight on two slots
    Z = S.copy()
    Z[(idx >> 3) & 1 == 1] = 0
    d_z = torch.from_numpy(C.to_mont(sf, Z).view(np.int64)).cuda()
    key.submit_pair_device(1, d_z, n, 3, is_mont=True)
    key.submit_pair_device(2, d_s, 1000, 2, is_mont=True)
    lo, hi = key.wait_pair(1)
    assert point_to_affine(c, hi) == (0, 0) and point_to_affine(c, lo) == C.jac_to_affine(c, C.msm_fast(c, B, Z))
    lo, hi = key.wait_pair(2)
    m = ((np.arange(1000) >> 2) & 1).astype(bool)
    a, b = S[:1000].copy(), S[:1000].copy()
    a[m] = 0
    b[~m] = 0
    assert point_to_affine(c, lo) == C.jac_to_affine(c, C.msm_fast(c, B[:1000], a)) and point_to_affine(c, hi) == C.jac_to_affine(c, C.msm_fast(c, B[:1000], b))
    # an ordinary commitment on the same slot afterwards; wait() on a pending pair is refused
    assert point_to_affine(c, key.commit(S)) == C.jac_to_affine(c, C.msm_fast(c, B, S))
    key.submit_pair_device(0, d_s, n, 5, is_mont=True)
    with pytest.raises(LurkHipError, match="pair"):
        key.wait(0)
    key.close()
    small = CommitmentKey(c, B[:5000], precompute=True)  # the small-commitment form has no key spaces
    assert not small.supports_pairs()
    with pytest.raises(LurkHipError):
        small.submit_pair_device(0, d_s, 5000, 1, is_mont=True)
    small.close()


DIRECT_CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
import lurk_beta_amd as L
from oracle import coracle as C
c = int(sys.argv[1])
sf = 1 if c == 0 else 0
q = (1 << 255)
outs = []
for log_n, plain in ((16, False), (14, False), (11, False), (6, False), (12, True)):
    n = 1 << log_n
    B = C.synth_bases(c, n).copy()
    B[3] = 0                                   # an identity base
    key = L.CommitmentKey(c, B, precompute=not plain, window_bits=0 if plain else 16)   # 16-bit windows: 32 768 buckets per key space
    assert key.info()["form"] == ("plain" if plain else "table")
    S = C.synth_scalars(sf, 90 + log_n, 0, n)
    S[: n // 3] = S[0]                         # a third of the scalars equal: every window has one bucket far above the per-lane cap
    S[n // 3] = 0
    S[n // 3 + 1 : n // 3 + 5] = C.ints_to_limbs([1, 2, (1 << 16) - 1, 1 << 15])
    want = C.jac_to_affine(c, C.msm_fast(c, B, S))
    got = L.point_to_affine(c, key.commit(S))
    assert got == want, (log_n, plain)
    outs.append(got)
    if not plain and log_n >= 11:
        d_s = torch.from_numpy(C.to_mont(sf, S).view(np.int64)).cuda()
        idx = np.arange(n)
        for bit in (0, log_n - 1):
            m = ((idx >> bit) & 1).astype(bool)
            lo_v, hi_v = S.copy(), S.copy()
            lo_v[m] = 0
            hi_v[~m] = 0
            key.submit_pair_device(0, d_s, n, bit, is_mont=True)
            lo, hi = key.wait_pair(0)
            assert L.point_to_affine(c, lo) == C.jac_to_affine(c, C.msm_fast(c, B, lo_v)), (log_n, bit)
            assert L.point_to_affine(c, hi) == C.jac_to_affine(c, C.msm_fast(c, B, hi_v)), (log_n, bit)
            outs.append(L.point_to_affine(c, lo))
        for k in range(3):                     # in flight on three slots
            key.submit_device(k, d_s, n - k, is_mont=True)
        for k in range(3):
            assert L.point_to_affine(c, key.wait(k)) == C.jac_to_affine(c, C.msm_fast(c, B[: n - k], S[: n - k])), (log_n, k)
    key.close()
print("digest", hash(tuple(outs)) & 0xffffffff)
print("child ok")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("cn,c", CURVES)
def test_few_bucket_commitments_take_the_direct_path(hip, cn, c):
    """msm_bucket_direct.hip: commitments with <= 2^17 buckets and <= 2^21 entries (a key of <= 2^16 points under 16-bit windows: the
    opening argument's folded key) sum each bucket with 1-8 adjacent lanes in one launch instead of plan + accumulate + finalize.  Against
    the oracle: 4 / 2 / 1 lanes per bucket (2^16 .. 2^6 points; pairs: 2), a third of the scalars equal (buckets handed to the workgroup
    kernel), zero / identity entries, pairs and commitments in flight, and a plain-form key (2^19 buckets: the planned-task stages);
    LURK_MSM_BUCKET_DIRECT=0 (the planned-task stages everywhere) must print the same points."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for sw in ("1", "0"):
        e = dict(os.environ)
        e["LURK_MSM_BUCKET_DIRECT"] = sw
        p = subprocess.run([sys.executable, "-c", DIRECT_CHILD % root, str(c)], env=e, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "child ok" in p.stdout
        digests.append([l for l in p.stdout.splitlines() if l.startswith("digest")][0])
    assert digests[0] == digests[1]

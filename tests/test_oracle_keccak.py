"""arecibo's Keccak256Transcript: the library's host implementation (lurk_hip_keccak_transcript_*) against the oracle's independent
restatement (oracle/keccak_transcript.py), and Keccak-256 itself against its known answers.  The transcript is restated from arecibo's
published source and UNPINNED (no transcript value exists in /root/reference); CPU only."""
import ctypes
import hashlib

import numpy as np

from oracle import coracle as C
from oracle import keccak_transcript as K
from oracle import pyref as R

KAT = {b"": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470",   # the empty-input Keccak-256 (Ethereum's empty hash)
       b"abc": "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"}


def _lib_keccak(data: bytes) -> bytes:
    from lurk_beta_amd import _lib

    out = ctypes.create_string_buffer(32)
    _lib.check(_lib.load().lurk_hip_keccak256(data, len(data), out))
    return out.raw


def test_keccak256_known_answers_and_the_permutation():
    for msg, want in KAT.items():
        assert K.keccak256(msg).hex() == want and _lib_keccak(msg).hex() == want
    rng = np.random.default_rng(3)
    for n in (1, 135, 136, 137, 271, 272, 1000):
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert _lib_keccak(m) == K.keccak256(m)

    # the oracle's Keccak-f is the one inside hashlib's SHA3-256 (same permutation, domain byte 0x06): an independent pin
    def sha3_via_oracle(data):
        rate = 136
        msg = bytearray(data) + b"\x00" * (rate - len(data) % rate)
        msg[len(data)] ^= 0x06
        msg[-1] ^= 0x80
        a = [[0] * 5 for _ in range(5)]
        for off in range(0, len(msg), rate):
            for i in range(rate // 8):
                a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
            a = K.keccak_f(a)
        return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))

    for m in (b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 500):
        assert sha3_via_oracle(m) == hashlib.sha3_256(m).digest()


def test_library_transcript_equals_the_oracles():
    from lurk_beta_amd import _lib
    from lurk_beta_amd.spartan import Transcript

    lib = _lib.load()
    for curve, f in ((0, 1), (1, 0)):
        q = R.modulus(f)
        cn = "pallas" if curve == 0 else "vesta"
        t_lib = Transcript(b"test", curve)
        t_orc = K.KeccakTranscript(b"lurk-hip spartan v2" + b"test")
        rng = np.random.default_rng(11 + curve)
        pts = [C.gen_mul(curve, 5), C.gen_mul(curve, 0), C.gen_mul(curve, 123456789)]  # the middle one is the identity
        for step in range(40):
            kind = step % 5
            if kind == 0:
                xs = [int(x) for x in C.limbs_to_ints(C.synth_scalars(f, 300 + step, 0, 1 + step % 7))] + [0, q - 1]
                t_lib.absorb_scalars(b"s", xs)
                t_orc.absorb_scalars(b"s", xs)
            elif kind == 1:
                jac = pts[step % 3]
                aff = C.jac_to_affine(curve, jac)
                t_lib.absorb_jacobian(b"pt", jac)
                t_orc.absorb_point(b"pt", aff)
                t_lib.absorb_point(b"pt2", aff if aff else (0, 0))
                t_orc.absorb_point(b"pt2", aff)
            elif kind == 2:
                raw = rng.integers(0, 256, 1 + 37 * step, dtype=np.uint8).tobytes()
                t_lib.absorb(b"raw", raw)
                t_orc.absorb(b"raw", raw)
            elif kind == 3:
                _lib.check(lib.lurk_hip_keccak_transcript_dom_sep(t_lib._h, b"sep", 3))
                t_orc.dom_sep(b"sep")
            else:
                for lab in (b"c", b"challenge", b""):
                    assert t_lib.squeeze(lab, q) == t_orc.squeeze(lab, q), (cn, step, lab)
        assert t_lib.squeeze(b"end", q) == t_orc.squeeze(b"end", q)


def test_round_bindings_equal_the_oracles_rounds():
    """lurk_hip_keccak_sumcheck_challenge / lurk_hip_keccak_ipa_challenge (the `challenge` arguments that keep a proof's rounds inside
    the library): round by round the oracle's transcript, absorbing the same coefficients / points under the same labels, squeezes the
    same challenges; the binding keeps them in order and refuses to overrun its buffer."""
    from lurk_beta_amd import _lib
    from lurk_beta_amd.spartan import Transcript

    lib = _lib.load()
    for curve, f in ((0, 1), (1, 0)):
        q = R.modulus(f)
        t_lib = Transcript(b"rounds", curve)
        t_orc = K.KeccakTranscript(b"lurk-hip spartan v2" + b"rounds")
        sc = t_lib.rounds(q, b"p", b"c", cap=5)
        sc.struct.n_scalars = 4
        fn_ptr, user = sc.callback("sumcheck")
        want = []
        for j in range(5):
            coeffs = C.limbs_to_ints(C.synth_scalars(f, 400 + j, 0, 4))
            arr = C.ints_to_limbs(coeffs)
            out = np.zeros(4, dtype=np.uint64)
            _lib.check(lib.lurk_hip_keccak_sumcheck_challenge(user, j, _lib.ptr(arr), _lib.ptr(out)))
            t_orc.absorb_scalars(b"p", coeffs)
            want.append(t_orc.squeeze(b"c", q))
            assert C.limbs_to_ints(out.reshape(1, 4))[0] == want[-1]
        assert sc.challenges() == want
        arr, out = C.ints_to_limbs([1, 2, 3, 4]), np.zeros(4, dtype=np.uint64)
        assert lib.lurk_hip_keccak_sumcheck_challenge(user, 5, _lib.ptr(arr), _lib.ptr(out)) != 0  # a sixth challenge does not fit
        ip = t_lib.rounds(q, b"L", b"r", absorb2=b"R", cap=4)
        t_orc2 = K.KeccakTranscript(b"lurk-hip spartan v2" + b"rounds2")
        t_lib2 = Transcript(b"rounds2", curve)
        ip = t_lib2.rounds(q, b"L", b"r", absorb2=b"R", cap=4)
        pts = [C.gen_mul(curve, 7), C.gen_mul(curve, 0), C.gen_mul(curve, 99), C.gen_mul(curve, 3)]
        got, want = [], []
        for j in range(3):
            L, Rr = pts[j], pts[j + 1]
            got.append(ip.ipa_round(j, L, Rr))
            t_orc2.absorb_point(b"L", C.jac_to_affine(curve, L))
            t_orc2.absorb_point(b"R", C.jac_to_affine(curve, Rr))
            want.append(t_orc2.squeeze(b"r", q))
        assert got == want == ip.challenges()
        assert fn_ptr and user


def test_from_uniform_reduction_edges():
    """Scalar::from_uniform on the extremes of the 512-bit range (through the squeeze's reduction, fed directly)."""
    from lurk_beta_amd import _lib

    # the reduction is private to the transcript; exercise it through squeezes of many labels and compare with the oracle's plain
    # `int.from_bytes(out, "little") % modulus` - 200 challenges per field, which all went through lo + hi * 2^256 mod p
    from lurk_beta_amd.spartan import Transcript

    for f in (0, 1, 2):
        q = R.modulus(f)
        t_lib, t_orc = Transcript(b"edge"), K.KeccakTranscript(b"lurk-hip spartan v2" + b"edge")
        for k in range(200):
            lab = b"l%d" % k
            a, b = t_lib.squeeze(lab, q), t_orc.squeeze(lab, q)
            assert a == b and 0 <= a < q

"""ORACLE (test infrastructure, never imported by the product): arecibo's Keccak256Transcript, the transcript of RelaxedR1CSSNARK /
BatchedRelaxedR1CSSNARK as CompressedSNARK::prove runs them (/root/reference/src/proof/nova.rs:92, 341-356;
/root/reference/src/proof/supernova.rs:110, 293-302).  arecibo is an un-vendored git dependency (/root/reference/Cargo.toml:128), so this
is RESTATED FROM ITS PUBLISHED SOURCE [MEM: src/provider/keccak.rs, traits/mod.rs, provider/pasta.rs, provider/pedersen.rs] and is
UNPINNED: the reference tree holds no transcript value.  What is pinned is Keccak-256 itself (the two known answers below).

    PERSONA_TAG = b"NoTR", DOM_SEP_TAG = b"NoDS"; state: 64 bytes; round: u16; a running Keccak256 hasher
    new(label):        state = updated_state(Keccak256(), PERSONA_TAG || label); round = 0; hasher = Keccak256()
    absorb(label, o):  hasher.update(label); hasher.update(o.to_transcript_bytes())
    dom_sep(bytes):    hasher.update(DOM_SEP_TAG); hasher.update(bytes)
    squeeze(label):    input = DOM_SEP_TAG || round (u16 LE) || state || label
                       out = updated_state(hasher, input); round += 1; state = out; hasher = Keccak256()
                       challenge = Scalar::from_uniform(out)          (64 bytes, little-endian integer mod the field order)
    updated_state(h, input): h.update(input); lo = h.clone().update([0]).finalize(); hi = h.update([1]).finalize(); lo || hi
    to_transcript_bytes: a field element = its 32-byte repr REVERSED (big-endian); a commitment = x || y || [1 if finite else 0]
    with (x, y) = (0, 0) for the identity; a slice = the concatenation of its elements' bytes.
"""
from __future__ import annotations

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_M = (1 << 64) - 1


def _rol(x, n):
    return ((x << n) | (x >> (64 - n))) & _M


def keccak_f(a):
    """Keccak-f[1600] on a[x][y] (FIPS 202, section 3.2), plain statement."""
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        # rho and pi together: lane (x, y) rotated by its offset lands at (y, 2x + 3y)
        offs = [[0] * 5 for _ in range(5)]
        xx, yy = 1, 0
        for t in range(24):
            offs[xx][yy] = ((t + 1) * (t + 2) // 2) % 64
            xx, yy = yy, (2 * xx + 3 * yy) % 5
        for xx in range(5):
            for yy in range(5):
                b[yy][(2 * xx + 3 * yy) % 5] = _rol(a[xx][yy], offs[xx][yy]) if offs[xx][yy] else a[xx][yy]
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & _M & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    return a


class Keccak256:
    """The pre-standard Keccak-256 (pad10*1 with domain byte 0x01, rate 136), incremental, clonable."""
    RATE = 136

    def __init__(self):
        self.buf = b""

    def update(self, data: bytes):
        self.buf += bytes(data)
        return self

    def clone(self):
        k = Keccak256()
        k.buf = self.buf
        return k

    def finalize(self) -> bytes:
        msg = bytearray(self.buf)
        pad = self.RATE - len(msg) % self.RATE
        msg += b"\x00" * pad
        msg[len(self.buf)] ^= 0x01
        msg[-1] ^= 0x80
        a = [[0] * 5 for _ in range(5)]
        for off in range(0, len(msg), self.RATE):
            for i in range(self.RATE // 8):
                a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
            a = keccak_f(a)
        return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def keccak256(data: bytes) -> bytes:
    return Keccak256().update(data).finalize()


PERSONA_TAG, DOM_SEP_TAG = b"NoTR", b"NoDS"


def _updated_state(h: Keccak256, data: bytes) -> bytes:
    h = h.clone().update(data)
    return h.clone().update(b"\x00").finalize() + h.clone().update(b"\x01").finalize()


def scalar_bytes(x: int) -> bytes:
    return int(x).to_bytes(32, "big")  # to_repr() reversed


def point_bytes(pt) -> bytes:
    """pt: affine (x, y) as integers, or None / (0, 0) for the identity."""
    if pt is None or tuple(pt) == (0, 0):
        return scalar_bytes(0) + scalar_bytes(0) + b"\x00"
    return scalar_bytes(pt[0]) + scalar_bytes(pt[1]) + b"\x01"


class KeccakTranscript:
    def __init__(self, label: bytes):
        self.round = 0
        self.state = _updated_state(Keccak256(), PERSONA_TAG + label)
        self.hasher = Keccak256()

    def absorb(self, label: bytes, data: bytes):
        self.hasher.update(label).update(data)

    def absorb_scalars(self, label: bytes, xs):
        self.absorb(label, b"".join(scalar_bytes(x) for x in xs))

    def absorb_point(self, label: bytes, pt):
        self.absorb(label, point_bytes(pt))

    def dom_sep(self, data: bytes):
        self.hasher.update(DOM_SEP_TAG).update(data)

    def squeeze(self, label: bytes, modulus: int) -> int:
        out = _updated_state(self.hasher, DOM_SEP_TAG + self.round.to_bytes(2, "little") + self.state + label)
        self.round += 1
        self.state = out
        self.hasher = Keccak256()
        return int.from_bytes(out, "little") % modulus

"""LURKDUMP - the files a Rust host writes so that the REAL step of benches/fibonacci.rs can be measured and checked here.

The synthetic R1CS and witness of `bench.py --workload fold_step` are a builder-chosen model ("replace W with
real dumped witnesses once a Rust host is available").  This module is the other half of `rust/lurk-hip-sys/src/dump.rs`: three
little-endian files written from arecibo's `R1CSShape`, the fresh `R1CSWitness` / instance of every folding step and the
`CommitmentKey` of a `PublicParams` (/root/reference/benches/fibonacci.rs:98-122, /root/reference/src/proof/nova.rs:196-216,
282-295, /root/reference/src/lem/multiframe.rs:856-920), read back as the arrays `R1CSShape`, `FoldingContext` and
`CommitmentKey` of this package take.

Every file starts with one 64-byte header:

    off  size  field
      0     8  magic   b"LURKDUMP"
      8     4  version u32 = 1
     12     4  kind    u32   1 = R1CS shape, 2 = witnesses, 3 = commitment key
     16     4  id      u32   shape / witnesses: LURK_FIELD_* of the scalar field (0 = Pallas Fp = vesta::Scalar,
                             1 = Pallas Fq = pallas::Scalar, 2 = BN254 Fr); key: LURK_CURVE_* (0 = Pallas, 1 = Vesta)
     20     4  encoding u32  0 = canonical little-endian integers (`PrimeField::to_repr()`);
                             1 = Montgomery limbs exactly as pasta_curves (feature repr-c) holds them in memory: a dump is a byte copy
     24     8  a       u64   shape: num_cons        witnesses: num_vars        key: npoints
     32     8  b       u64   shape: num_vars        witnesses: num_io          key: 0
     40     8  c       u64   shape: num_io          witnesses: steps           key: 0
     48    16  reserved (zero)

Bodies (field elements are 32 bytes, 4 x u64 little-endian, in the header's encoding):

    shape      for M in A, B, C (arecibo `SparseMatrix`, CSR over z = [W | u | X]: column < num_vars is W, == num_vars is u, above is X):
                   nnz u64, indptr (num_cons + 1) x u64, indices nnz x u64, data nnz x 32 B
    witnesses  pp_digest 32 B (always canonical), then per step: W num_vars x 32 B, X num_io x 32 B
    key        npoints x 64 B affine (x, y); the identity is (0, 0)

Host-side plumbing only (numpy): nothing here computes on field elements; a canonical dump is brought to Montgomery form on the
device by `to_montgomery_device` (one fold_vec launch).
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"LURKDUMP"
VERSION = 1
KIND_SHAPE, KIND_WITNESS, KIND_KEY = 1, 2, 3
ENC_CANONICAL, ENC_MONTGOMERY = 0, 1
HEADER_BYTES = 64
_HDR = struct.Struct("<8sIIIIQQQ16x")
assert _HDR.size == HEADER_BYTES

# R^2 mod p for R = 2^256, as plain integers: the multiplier that takes a canonical element to Montgomery form in one Montgomery product
_MODULUS = {
    0: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,  # Pallas Fp
    1: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,  # Pallas Fq
    2: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,  # BN254 Fr
}


def _header(kind, ident, encoding, a, b, c):
    return _HDR.pack(MAGIC, VERSION, kind, ident, encoding, a, b, c)


def _read_header(fh, want_kind):
    raw = fh.read(HEADER_BYTES)
    if len(raw) != HEADER_BYTES:
        raise ValueError("LURKDUMP: truncated header")
    magic, version, kind, ident, encoding, a, b, c = _HDR.unpack(raw)
    if magic != MAGIC:
        raise ValueError("LURKDUMP: bad magic")
    if version != VERSION:
        raise ValueError(f"LURKDUMP: version {version}, this reader knows {VERSION}")
    if kind != want_kind:
        raise ValueError(f"LURKDUMP: kind {kind}, expected {want_kind}")
    if encoding not in (ENC_CANONICAL, ENC_MONTGOMERY):
        raise ValueError(f"LURKDUMP: unknown encoding {encoding}")
    return ident, encoding, a, b, c


def _take(fh, dtype, count, width=1):
    arr = np.fromfile(fh, dtype=dtype, count=count * width)
    if arr.size != count * width:
        raise ValueError("LURKDUMP: truncated body")
    return arr.reshape(count, width) if width > 1 else arr


def write_shape(path, field_id, num_cons, num_vars, num_io, mats, encoding=ENC_MONTGOMERY):
    """mats: (indptr, indices, data) for A, B, C; data n x 4 uint64 in `encoding`"""
    with open(path, "wb") as fh:
        fh.write(_header(KIND_SHAPE, field_id, encoding, num_cons, num_vars, num_io))
        for indptr, indices, data in mats:
            indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
            indices = np.ascontiguousarray(indices, dtype=np.uint64)
            data = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4)
            if indptr.size != num_cons + 1 or int(indptr[-1]) != indices.size or data.shape[0] != indices.size:
                raise ValueError("write_shape: inconsistent CSR arrays")
            fh.write(struct.pack("<Q", indices.size))
            indptr.tofile(fh)
            indices.tofile(fh)
            data.tofile(fh)


def read_shape(path):
    """-> dict(field_id, encoding, num_cons, num_vars, num_io, mats=[(indptr, indices, data) x 3])"""
    with open(path, "rb") as fh:
        field_id, encoding, num_cons, num_vars, num_io = _read_header(fh, KIND_SHAPE)
        mats = []
        for _ in range(3):
            (nnz,) = struct.unpack("<Q", fh.read(8))
            indptr = _take(fh, np.uint64, num_cons + 1)
            indices = _take(fh, np.uint64, nnz)
            data = _take(fh, np.uint64, nnz, 4)
            if int(indptr[0]) != 0 or int(indptr[-1]) != nnz or (nnz and int(indices.max()) > num_vars + num_io):
                raise ValueError("LURKDUMP: CSR arrays out of range")
            mats.append((indptr, indices, data))
        if fh.read(1):
            raise ValueError("LURKDUMP: trailing bytes after the shape")
    return {"field_id": field_id, "encoding": encoding, "num_cons": num_cons, "num_vars": num_vars, "num_io": num_io, "mats": mats}


def write_witnesses(path, field_id, pp_digest, steps, encoding=ENC_MONTGOMERY):
    """steps: [(W num_vars x 4 uint64, X num_io x 4 uint64), ...] in `encoding`; pp_digest: a Python int (canonical)"""
    steps = [(np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4), np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)) for w, x in steps]
    if not steps:
        raise ValueError("write_witnesses: no steps")
    num_vars, num_io = steps[0][0].shape[0], steps[0][1].shape[0]
    with open(path, "wb") as fh:
        fh.write(_header(KIND_WITNESS, field_id, encoding, num_vars, num_io, len(steps)))
        fh.write(int(pp_digest).to_bytes(32, "little"))
        for w, x in steps:
            if w.shape[0] != num_vars or x.shape[0] != num_io:
                raise ValueError("write_witnesses: steps of different sizes")
            w.tofile(fh)
            x.tofile(fh)


def read_witnesses(path):
    """-> dict(field_id, encoding, num_vars, num_io, pp_digest (int), steps=[(W, X), ...])"""
    with open(path, "rb") as fh:
        field_id, encoding, num_vars, num_io, count = _read_header(fh, KIND_WITNESS)
        digest = fh.read(32)
        if len(digest) != 32:
            raise ValueError("LURKDUMP: truncated body")
        steps = [(_take(fh, np.uint64, num_vars, 4), _take(fh, np.uint64, num_io, 4) if num_io else np.zeros((0, 4), dtype=np.uint64))
                 for _ in range(count)]
        if fh.read(1):
            raise ValueError("LURKDUMP: trailing bytes after the witnesses")
    return {"field_id": field_id, "encoding": encoding, "num_vars": num_vars, "num_io": num_io, "pp_digest": int.from_bytes(digest, "little"),
            "steps": steps}


def write_key(path, curve, points, encoding=ENC_MONTGOMERY):
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    with open(path, "wb") as fh:
        fh.write(_header(KIND_KEY, curve, encoding, points.shape[0], 0, 0))
        points.tofile(fh)


def read_key(path):
    """-> dict(curve, encoding, points n x 8 uint64)"""
    with open(path, "rb") as fh:
        curve, encoding, n, _, _ = _read_header(fh, KIND_KEY)
        pts = _take(fh, np.uint64, n, 8)
        if fh.read(1):
            raise ValueError("LURKDUMP: trailing bytes after the key")
    return {"curve": curve, "encoding": encoding, "points": pts}


def r2_limbs(field_id):
    """R^2 mod p (R = 2^256) as 4 uint64 limbs: one Montgomery product by it takes canonical limbs to Montgomery limbs"""
    p = _MODULUS[field_id]
    v = pow(1 << 256, 2, p)
    return np.array([(v >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)


def to_montgomery_device(field_id, d_x):
    """canonical elements resident on the device (torch int64, n x 4) -> their Montgomery form, by the library's own fold kernel:
    0 + (R^2) (x) x with (x) the Montgomery product = x R mod p"""
    import torch

    from . import fold_vec

    return fold_vec(field_id, torch.zeros_like(d_x), d_x, r2_limbs(field_id).reshape(1, 4))

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
     12     4  kind    u32   1 = R1CS shape, 2 = witnesses, 3 = commitment key, 4 = probe (round 6)
     16     4  id      u32   shape / witnesses: LURK_FIELD_* of the scalar field (0 = Pallas Fp = vesta::Scalar,
                             1 = Pallas Fq = pallas::Scalar, 2 = BN254 Fr); key / probe: LURK_CURVE_* (0 = Pallas, 1 = Vesta)
     20     4  encoding u32  0 = canonical little-endian integers (`PrimeField::to_repr()`);
                             1 = Montgomery limbs exactly as pasta_curves (feature repr-c) holds them in memory: a dump is a byte copy
     24     8  a       u64   shape: num_cons        witnesses: num_vars        key: npoints      probe: num_io
     32     8  b       u64   shape: num_vars        witnesses: num_io          key: 0            probe: absorbed elements (0 = none)
     40     8  c       u64   shape: num_io          witnesses: steps           key: 0            probe: key points (0 = none)
     48    16  reserved (zero)

Bodies (field elements are 32 bytes, 4 x u64 little-endian, in the header's encoding):

    shape      for M in A, B, C (arecibo `SparseMatrix`, CSR over z = [W | u | X]: column < num_vars is W, == num_vars is u, above is X):
                   nnz u64, indptr (num_cons + 1) x u64, indices nnz x u64, data nnz x 32 B
    witnesses  pp_digest 32 B (always canonical), then per step: W num_vars x 32 B, X num_io x 32 B
    key        npoints x 64 B affine (x, y); the identity is (0, 0)
    probe      ONE (transcript inputs -> r) pair of a real NIFS::prove and the first points of the real key - what localises a mismatch
               of the two restatements that were written from memory (transcript.hip, keygen.hip) in minutes instead of a rebuild:
                   pp_digest 32 B (always canonical)
                   U1: comm_W 64 B, comm_E 64 B (affine), u 32 B, X num_io x 32 B     U2: comm_W 64 B, X num_io x 32 B     comm_T 64 B
                   r 32 B (the challenge NIFS::prove squeezed, an element of the scalar field)
                   absorbed b x 32 B: the elements arecibo's RO held when it squeezed (`ro.state`; elements of the OTHER field of the cycle)
                   label: length u64, then the bytes zero-padded to a multiple of 8     key points c x 64 B affine: ck[0..c) of from_label(label)
               `python -m lurk_beta_amd.dump probe FILE` reports which stage disagrees and `--search` walks lurk_hip_ro_params /
               lurk_hip_ck_params (lurk_beta_amd/params.py) until the record is reproduced.

Host-side plumbing only (numpy): nothing here computes on field elements; a canonical dump is brought to Montgomery form on the
device by `to_montgomery_device` (one fold_vec launch).
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"LURKDUMP"
VERSION = 1
KIND_SHAPE, KIND_WITNESS, KIND_KEY, KIND_PROBE = 1, 2, 3, 4
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


# ---- kind 4: the probe record (round 6) -------------------------------------------------------------------------------------------
_PROBE_POINTS = ("comm_W1", "comm_E1", "comm_W2", "comm_T")


def write_probe(path, curve, num_io, pp_digest, comm_W1, comm_E1, u1, x1, comm_W2, x2, comm_T, r, absorbed=None, label=b"", key_points=None,
                encoding=ENC_MONTGOMERY):
    """Points: 8 uint64 (affine x, y; identity (0, 0)); u1 / r: 4 uint64; x1 / x2: num_io x 4; absorbed: n x 4 (elements of the curve's BASE
    field) or None; key_points: c x 8 or None - all in `encoding`; pp_digest: a Python int (canonical)."""
    a = lambda v, w: np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, w)
    absorbed = np.zeros((0, 4), dtype=np.uint64) if absorbed is None else a(absorbed, 4)
    key_points = np.zeros((0, 8), dtype=np.uint64) if key_points is None else a(key_points, 8)
    x1, x2 = a(x1, 4), a(x2, 4)
    if x1.shape[0] != num_io or x2.shape[0] != num_io:
        raise ValueError("write_probe: X1 / X2 do not hold num_io elements")
    label = bytes(label)
    with open(path, "wb") as fh:
        fh.write(_header(KIND_PROBE, curve, encoding, num_io, absorbed.shape[0], key_points.shape[0]))
        fh.write(int(pp_digest).to_bytes(32, "little"))
        for part, width in ((comm_W1, 8), (comm_E1, 8), (u1, 4), (x1, 4), (comm_W2, 8), (x2, 4), (comm_T, 8), (r, 4), (absorbed, 4)):
            a(part, width).tofile(fh)
        fh.write(struct.pack("<Q", len(label)))
        fh.write(label + bytes(-len(label) % 8))
        key_points.tofile(fh)


def read_probe(path):
    """-> dict(curve, encoding, num_io, pp_digest (int), comm_W1, comm_E1, u1, x1, comm_W2, x2, comm_T, r, absorbed, label, key_points)"""
    with open(path, "rb") as fh:
        curve, encoding, num_io, n_abs, n_key = _read_header(fh, KIND_PROBE)
        if curve not in (0, 1):
            raise ValueError(f"LURKDUMP: probe for unknown curve {curve}")
        digest = fh.read(32)
        if len(digest) != 32:
            raise ValueError("LURKDUMP: truncated body")
        out = {"curve": curve, "encoding": encoding, "num_io": num_io, "pp_digest": int.from_bytes(digest, "little")}
        for name, count, width in (("comm_W1", 1, 8), ("comm_E1", 1, 8), ("u1", 1, 4), ("x1", num_io, 4), ("comm_W2", 1, 8), ("x2", num_io, 4),
                                   ("comm_T", 1, 8), ("r", 1, 4), ("absorbed", n_abs, 4)):
            out[name] = _take(fh, np.uint64, count, width) if count else np.zeros((0, width), dtype=np.uint64)
        raw = fh.read(8)
        if len(raw) != 8:
            raise ValueError("LURKDUMP: truncated body")
        (llen,) = struct.unpack("<Q", raw)
        if llen > 1 << 16:
            raise ValueError("LURKDUMP: label length out of range")
        lab = fh.read(llen + (-llen % 8))
        if len(lab) != llen + (-llen % 8):
            raise ValueError("LURKDUMP: truncated body")
        out["label"] = lab[:llen]
        out["key_points"] = _take(fh, np.uint64, n_key, 8) if n_key else np.zeros((0, 8), dtype=np.uint64)
        if fh.read(1):
            raise ValueError("LURKDUMP: trailing bytes after the probe")
    return out


def _host_mont(field_id, arr, encoding, to_mont):
    """n x 4 uint64 between canonical and Montgomery form on the HOST (Python integers: a probe holds a few dozen elements)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    if (encoding == ENC_MONTGOMERY) == to_mont:
        return arr.copy()
    p = _MODULUS[field_id]
    f = (1 << 256) % p if to_mont else pow(1 << 256, -1, p)
    out = np.zeros_like(arr)
    for i, row in enumerate(arr):
        v = sum(int(row[k]) << (64 * k) for k in range(4)) * f % p
        out[i] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    return out


def _probe_inputs(pr):
    """The probe's transcript inputs as lurk_hip_nifs_challenge takes them: Montgomery scalars, 96-byte Jacobians (z = 1; identity z = 0)."""
    curve = pr["curve"]
    sf, bf = 1 - curve, curve  # scalar / base field ids: Pallas has scalars in Fq (1) and coordinates in Fp (0)
    one = _host_mont(bf, np.array([[1, 0, 0, 0]], dtype=np.uint64), ENC_CANONICAL, True)[0]

    def jac(pt):
        xy = _host_mont(bf, np.asarray(pt).reshape(2, 4), pr["encoding"], True)
        if not xy.any():
            return np.zeros(12, dtype=np.uint64)
        return np.concatenate([xy.reshape(-1), one])

    m = lambda v: _host_mont(sf, v, pr["encoding"], True)
    return (curve, pr["pp_digest"], jac(pr["comm_W1"]), jac(pr["comm_E1"]), m(pr["u1"]), m(pr["x1"]), jac(pr["comm_W2"]), m(pr["x2"]), jac(pr["comm_T"]))


def _ints(arr):
    return [sum(int(r[k]) << (64 * k) for k in range(4)) for r in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


def check_probe(pr):
    """Run the library's transcript and key generation (host code, the parameters in force) over a probe record.  -> dict of stage ->
    True / False / None (None: the record does not carry what the stage needs), with the first disagreeing position where there is one."""
    from . import nifs_challenge, nova_ro_squeeze
    from . import params as P

    curve = pr["curve"]
    sf, bf = 1 - curve, curve
    args = _probe_inputs(pr)
    want_r = _ints(_host_mont(sf, pr["r"], pr["encoding"], False))[0]
    got_r = _ints(_host_mont(sf, nifs_challenge(*args).reshape(1, 4), ENC_MONTGOMERY, False))[0]
    out = {"r": got_r == want_r, "absorb_list": None, "sponge": None, "key": None}
    if len(pr["absorbed"]):
        theirs = _ints(_host_mont(bf, pr["absorbed"], pr["encoding"], False))
        ours = P.nifs_absorb_list(*args)
        out["absorb_list"] = ours == theirs
        if ours != theirs:
            out["absorb_list_first_difference"] = next((i for i, (x, y) in enumerate(zip(ours, theirs)) if x != y), min(len(ours), len(theirs)))
            out["absorb_list_lengths"] = (len(ours), len(theirs))
        bits = P.ro_params_get()["num_challenge_bits"]
        out["sponge"] = nova_ro_squeeze(bf, theirs, bits) % _MODULUS[sf] == want_r
    if len(pr["key_points"]):
        ours = P.ck_from_label_host(curve, pr["label"], len(pr["key_points"]))
        theirs = _host_mont(bf, pr["key_points"].reshape(-1, 4), pr["encoding"], True).reshape(-1, 8)
        out["key"] = bool(np.array_equal(ours, theirs))
        if not out["key"]:
            out["key_first_difference"] = int(np.nonzero((ours != theirs).any(axis=1))[0][0])
    return out


def search_probe(pr, limit=None):
    """Walk the parameter blocks until the library reproduces the record.  -> {"ro": moves or None, "ck": moves or None} where moves is the
    dict of fields that differ from the defaults ({} = the defaults already match).  The parameters in force are left as they were."""
    import itertools

    from . import nifs_challenge, nova_ro_squeeze
    from . import params as P

    curve = pr["curve"]
    sf, bf = 1 - curve, curve
    args = _probe_inputs(pr)
    want_r = _ints(_host_mont(sf, pr["r"], pr["encoding"], False))[0]
    found = {"ro": None, "ck": None}
    before_ro, before_ck = P.ro_params_get(), P.ck_params_get()
    try:
        defaults = P.ro_params_set()
        perms4 = [list(p) for p in itertools.permutations(range(4))]
        list_moves = [dict(item_order=io, relaxed_order=ro, fresh_order=fo, point_elements=pe, relaxed_x_limbs=rl, fresh_x_limbs=fl)
                      for pe in (3, 2) for rl in (4, 0) for fl in (0, 4) for fo in ([0, 1], [1, 0]) for io in perms4 for ro in perms4]
        n_theirs = len(pr["absorbed"])
        sponge_moves = [dict(arity=ar, pattern_absorbs=pa, squeeze_element=sq, num_challenge_bits=nb, domain_separator=ds)
                        for nb in (128, 250, 127) for sq in (0, 1) for ds in (0, 1) for ar in (24, 25, 16, 12, 8, 4, 2) for pa in sorted({0, 9, 19, 24, n_theirs})]
        diff = lambda mv: {k: v for k, v in mv.items() if defaults[k] != v}
        # fewest moved fields first: the nearest explanation of a mismatch is the one reported
        list_moves.sort(key=lambda mv: len(diff(mv)))
        sponge_moves.sort(key=lambda mv: len(diff(mv)))
        tried = 0
        if n_theirs:  # two independent searches: the list against the list, the sponge over THEIR list against r
            theirs = _ints(_host_mont(bf, pr["absorbed"], pr["encoding"], False))
            lm = None
            for mv in list_moves:
                P.ro_params_set(defaults, **mv)
                tried += 1
                if P.nifs_absorb_list(*args) == theirs:
                    lm = mv
                    break
            sm = None
            P.ro_params_set(defaults)
            for mv in sponge_moves:
                P.ro_params_set(defaults, **mv)
                tried += 1
                if nova_ro_squeeze(bf, theirs, mv["num_challenge_bits"]) % _MODULUS[sf] == want_r:
                    sm = mv
                    break
            if lm is not None and sm is not None:
                found["ro"] = diff({**lm, **sm})
        else:  # r alone: list and sponge moves together, by the total number of moved fields
            by_l, by_s = {}, {}
            for mv in list_moves:
                by_l.setdefault(len(diff(mv)), []).append(mv)
            for mv in sponge_moves:
                by_s.setdefault(len(diff(mv)), []).append(mv)
            for total in range(0, max(by_l) + max(by_s) + 1):
                for dl in range(0, total + 1):
                    for lm in by_l.get(dl, []):
                        for sm in by_s.get(total - dl, []):
                            P.ro_params_set(defaults, **lm, **sm)
                            tried += 1
                            got = _ints(_host_mont(sf, nifs_challenge(*args).reshape(1, 4), ENC_MONTGOMERY, False))[0]
                            if got == want_r:
                                found["ro"] = diff({**lm, **sm})
                                break
                            if limit and tried >= limit:
                                break
                        if found["ro"] is not None or (limit and tried >= limit):
                            break
                    if found["ro"] is not None or (limit and tried >= limit):
                        break
                if found["ro"] is not None or (limit and tried >= limit):
                    break
        found["ro_tried"] = tried
        if len(pr["key_points"]):
            ck_defaults = P.ck_params_set()
            theirs = _host_mont(bf, pr["key_points"].reshape(-1, 4), pr["encoding"], True).reshape(-1, 8)
            first = theirs[:1]
            for xof in (0, 1):
                for bpp in (32, 64, 48, 16):
                    for prefix in ("from_uniform_bytes", "from_uniform_bytes_", "ck", pr["label"].decode("latin1")):
                        for suite in ("_XMD:BLAKE2b_SSWU_RO_", "_XMD:BLAKE2b_SSWU_NU_"):
                            mv = dict(xof=xof, bytes_per_point=bpp, domain_prefix=prefix, suite=suite)
                            try:
                                P.ck_params_set(ck_defaults, **mv)
                            except Exception:  # noqa: BLE001 (a combination the library refuses: too long for two BLAKE2b blocks)
                                continue
                            if np.array_equal(P.ck_from_label_host(curve, pr["label"], 1), first) and \
                                    np.array_equal(P.ck_from_label_host(curve, pr["label"], len(theirs)), theirs):
                                found["ck"] = {k: v for k, v in mv.items() if ck_defaults[k] != v}
                                break
                        if found["ck"] is not None:
                            break
                    if found["ck"] is not None:
                        break
                if found["ck"] is not None:
                    break
    finally:
        P.ro_params_set(before_ro)
        P.ck_params_set(before_ck)
    return found


def _main(argv):
    import json

    if len(argv) < 2 or argv[0] != "probe":
        print("usage: python -m lurk_beta_amd.dump probe FILE [--search]")
        return 2
    pr = read_probe(argv[1])
    res = check_probe(pr)
    print(json.dumps({"check": res}, default=str))
    ok = all(v is not False for k, v in res.items() if k in ("r", "absorb_list", "sponge", "key"))
    if "--search" in argv and not ok:
        print(json.dumps({"search": search_probe(pr)}, default=str))
    return 0 if ok else 1


if __name__ == "__main__":
    import sys

    sys.exit(_main(sys.argv[1:]))

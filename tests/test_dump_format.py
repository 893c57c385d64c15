"""LURKDUMP (lurk_beta_amd/dump.py <-> rust/lurk-hip-sys/src/dump.rs): the files a Rust host writes so that the real fibonacci step can be
measured.  CPU: a synthetic shape / witness set / key round-trips through the format byte for byte, malformed files are refused, and the
Rust writer's constants are the reader's.  GPU (test_gpu_step.py::test_step_from_dump_files): a step driven from such files = the oracle."""
import os
import re
import struct

import numpy as np
import pytest

from lurk_beta_amd import dump

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _synthetic(seed=5, nc=37, nv=29, nio=2, steps=3):
    rng = np.random.default_rng(seed)

    def mat():
        cnt = rng.integers(0, 5, nc).astype(np.uint64)
        indptr = np.zeros(nc + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        return indptr, rng.integers(0, nv + 1 + nio, nnz).astype(np.uint64), rng.integers(0, 1 << 62, (nnz, 4), dtype=np.uint64)

    mats = [mat(), mat(), mat()]
    wit = [(rng.integers(0, 1 << 62, (nv, 4), dtype=np.uint64), rng.integers(0, 1 << 62, (nio, 4), dtype=np.uint64)) for _ in range(steps)]
    key = rng.integers(0, 1 << 62, (max(nc, nv), 8), dtype=np.uint64)
    return nc, nv, nio, mats, wit, key


@pytest.mark.parametrize("encoding", [dump.ENC_CANONICAL, dump.ENC_MONTGOMERY])
def test_round_trip(tmp_path, encoding):
    nc, nv, nio, mats, wit, key = _synthetic()
    ps, pw, pk = (str(tmp_path / n) for n in ("s.lurkdump", "w.lurkdump", "k.lurkdump"))
    dump.write_shape(ps, 1, nc, nv, nio, mats, encoding)
    dump.write_witnesses(pw, 1, 0xABCDEF0123456789, wit, encoding)
    dump.write_key(pk, 0, key, encoding)
    s = dump.read_shape(ps)
    assert (s["field_id"], s["encoding"], s["num_cons"], s["num_vars"], s["num_io"]) == (1, encoding, nc, nv, nio)
    for (a, b, c), (x, y, z) in zip(mats, s["mats"]):
        assert np.array_equal(a, x) and np.array_equal(b, y) and np.array_equal(c, z)
    w = dump.read_witnesses(pw)
    assert (w["field_id"], w["num_vars"], w["num_io"], w["pp_digest"], len(w["steps"])) == (1, nv, nio, 0xABCDEF0123456789, len(wit))
    for (a, b), (x, y) in zip(wit, w["steps"]):
        assert np.array_equal(a, x) and np.array_equal(b, y)
    k = dump.read_key(pk)
    assert k["curve"] == 0 and np.array_equal(k["points"], key)
    # the byte layout itself (what the Rust writer must produce): header fields at their offsets, first body word
    raw = open(ps, "rb").read()
    assert raw[:8] == b"LURKDUMP" and struct.unpack_from("<IIII", raw, 8) == (1, dump.KIND_SHAPE, 1, encoding)
    assert struct.unpack_from("<QQQ", raw, 24) == (nc, nv, nio) and raw[48:64] == bytes(16)
    assert struct.unpack_from("<Q", raw, 64)[0] == int(mats[0][0][-1])
    assert os.path.getsize(pw) == 64 + 32 + len(wit) * (nv + nio) * 32 and os.path.getsize(pk) == 64 + key.shape[0] * 64


def test_zero_io_and_empty_rows(tmp_path):
    nc, nv, nio, mats, wit, _ = _synthetic(seed=9, nio=0, steps=1)
    ps, pw = str(tmp_path / "s"), str(tmp_path / "w")
    dump.write_shape(ps, 0, nc, nv, 0, mats)
    dump.write_witnesses(pw, 0, 7, wit)
    assert dump.read_shape(ps)["num_io"] == 0
    w = dump.read_witnesses(pw)
    assert w["steps"][0][1].shape == (0, 4) and np.array_equal(w["steps"][0][0], wit[0][0])


def test_malformed_files_are_refused(tmp_path):
    nc, nv, nio, mats, wit, key = _synthetic()
    ps = str(tmp_path / "s")
    dump.write_shape(ps, 1, nc, nv, nio, mats)
    raw = open(ps, "rb").read()
    cases = {"magic": b"LURKDUMQ" + raw[8:], "version": raw[:8] + struct.pack("<I", 2) + raw[12:], "truncated": raw[:-5], "trailing": raw + b"\0",
             "kind": raw[:12] + struct.pack("<I", dump.KIND_KEY) + raw[16:], "encoding": raw[:20] + struct.pack("<I", 9) + raw[24:]}
    for name, blob in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        with pytest.raises(ValueError):
            dump.read_shape(p)
    bad = list(mats)
    bad[1] = (mats[1][0], mats[1][1][:-1], mats[1][2])
    with pytest.raises(ValueError):
        dump.write_shape(str(tmp_path / "bad"), 1, nc, nv, nio, bad)
    with pytest.raises(ValueError):  # a column beyond [W | u | X]
        far = (mats[0][0], np.full_like(mats[0][1], nv + nio + 1), mats[0][2])
        dump.write_shape(str(tmp_path / "far"), 1, nc, nv, nio, [far, mats[1], mats[2]])
        dump.read_shape(str(tmp_path / "far"))


def test_rust_writer_constants_are_the_readers():
    """rust/lurk-hip-sys/src/dump.rs cannot be compiled here; its constants and the order of the header fields are held to the reader's."""
    rs = open(os.path.join(ROOT, "rust", "lurk-hip-sys", "src", "dump.rs")).read()
    consts = dict(re.findall(r"pub const (\w+): (?:u32|usize) = (\d+);", rs))
    want = {"VERSION": dump.VERSION, "KIND_SHAPE": dump.KIND_SHAPE, "KIND_WITNESS": dump.KIND_WITNESS, "KIND_KEY": dump.KIND_KEY,
            "ENC_CANONICAL": dump.ENC_CANONICAL, "ENC_MONTGOMERY": dump.ENC_MONTGOMERY, "HEADER_BYTES": dump.HEADER_BYTES}
    assert {k: int(v) for k, v in consts.items()} == want
    assert 'pub const MAGIC: &[u8; 8] = b"LURKDUMP";' in rs and dump.MAGIC == b"LURKDUMP"
    hdr = rs[rs.index("fn header("):rs.index("/// The bytes of a slice")]
    assert hdr.index("MAGIC") < hdr.index("[VERSION, kind, id, encoding]") < hdr.index("[a, b, c]") < hdr.index("[0u8; 16]")
    assert "pub mod dump;" in open(os.path.join(ROOT, "rust", "lurk-hip-sys", "src", "lib.rs")).read()


def test_r2_constant():
    for f, p in dump._MODULUS.items():
        v = sum(int(x) << (64 * i) for i, x in enumerate(dump.r2_limbs(f)))
        assert v == pow(2, 512, p)

"""Run-time parameters of the two restatements that decide interoperability with arecibo: the Nova random oracle
(``lurk_hip_ro_params``) and commitment-key generation (``lurk_hip_ck_params``) - include/lurk_hip.h, round 6.

arecibo, neptune and pasta_curves are un-vendored dependencies of the reference (/root/reference/Cargo.toml:127-131) and
/root/reference holds no transcript value and no key bytes, so every constant that was recalled from memory is a field of these
two process-wide blocks (defaults = what rounds 1-5 compiled in) instead of a literal: the first Rust-side run whose ``r`` or
``ck[0]`` differs from arecibo's moves a field (or lets ``lurk_beta_amd.dump`` search them against a probe record) - no rebuild.
Host-side plumbing: nothing here needs a device."""
from __future__ import annotations

import contextlib
import ctypes

import numpy as np

from . import _lib

RO_ITEM_PP_DIGEST, RO_ITEM_U1, RO_ITEM_U2, RO_ITEM_COMM_T = 0, 1, 2, 3
RO_PART_COMM_W, RO_PART_COMM_E, RO_PART_U, RO_PART_X = 0, 1, 2, 3
CK_XOF_SHAKE256, CK_XOF_SHAKE128 = 0, 1

_RO_LISTS = {"item_order": 4, "relaxed_order": 4, "fresh_order": 2}
_CK_STRINGS = ("domain_prefix", "curve_name_pallas", "curve_name_vesta", "suite")


def _ro_to_dict(s: _lib.RoParamsStruct) -> dict:
    d = {}
    for name, _t in s._fields_:
        if name == "struct_size":
            continue
        v = getattr(s, name)
        d[name] = [int(x) for x in v] if name in _RO_LISTS else int(v)
    return d


def ro_params_get() -> dict:
    s = _lib.RoParamsStruct()
    _lib.check(_lib.load().lurk_hip_ro_params_get(ctypes.byref(s)))
    assert s.struct_size == ctypes.sizeof(_lib.RoParamsStruct)
    return _ro_to_dict(s)


def ro_params_set(params: dict | None = None, **changes) -> dict:
    """``ro_params_set()`` restores the defaults; otherwise the current block with ``params`` / keyword changes applied.  Returns the
    block that is now in force."""
    lib = _lib.load()
    if params is None and not changes:
        _lib.check(lib.lurk_hip_ro_params_set(None))
        return ro_params_get()
    cur = ro_params_get()
    cur.update(params or {})
    cur.update(changes)
    s = _lib.RoParamsStruct()
    s.struct_size = ctypes.sizeof(_lib.RoParamsStruct)
    for name, _t in s._fields_:
        if name == "struct_size":
            continue
        if name in _RO_LISTS:
            v = list(cur[name])
            if len(v) != _RO_LISTS[name]:
                raise ValueError(f"{name} takes {_RO_LISTS[name]} entries")
            setattr(s, name, (ctypes.c_uint32 * len(v))(*v))
        else:
            setattr(s, name, int(cur[name]))
    _lib.check(lib.lurk_hip_ro_params_set(ctypes.byref(s)))
    return ro_params_get()


@contextlib.contextmanager
def ro_params(**changes):
    """``with ro_params(point_elements=2): ...`` - the change holds inside the block, the previous block comes back after it."""
    before = ro_params_get()
    try:
        yield ro_params_set(**changes)
    finally:
        ro_params_set(before)


def ck_params_get() -> dict:
    s = _lib.CkParamsStruct()
    _lib.check(_lib.load().lurk_hip_ck_params_get(ctypes.byref(s)))
    assert s.struct_size == ctypes.sizeof(_lib.CkParamsStruct)
    d = {"xof": int(s.xof), "bytes_per_point": int(s.bytes_per_point)}
    for k in _CK_STRINGS:
        d[k] = bytes(getattr(s, k)).split(b"\0", 1)[0].decode()
    return d


def ck_params_set(params: dict | None = None, **changes) -> dict:
    lib = _lib.load()
    if params is None and not changes:
        _lib.check(lib.lurk_hip_ck_params_set(None))
        return ck_params_get()
    cur = ck_params_get()
    cur.update(params or {})
    cur.update(changes)
    s = _lib.CkParamsStruct()
    s.struct_size = ctypes.sizeof(_lib.CkParamsStruct)
    s.xof, s.bytes_per_point, s.reserved = int(cur["xof"]), int(cur["bytes_per_point"]), 0
    for k in _CK_STRINGS:
        raw = cur[k].encode()
        cap = ctypes.sizeof(dict(_lib.CkParamsStruct._fields_)[k])
        if len(raw) >= cap:
            raise ValueError(f"{k}: at most {cap - 1} bytes")
        setattr(s, k, raw)
    _lib.check(lib.lurk_hip_ck_params_set(ctypes.byref(s)))
    return ck_params_get()


@contextlib.contextmanager
def ck_params(**changes):
    before = ck_params_get()
    try:
        yield ck_params_set(**changes)
    finally:
        ck_params_set(before)


def ck_from_label_host(curve: int, label: bytes, n: int) -> np.ndarray:
    """The first ``n`` points of ``from_label(label)`` mapped on the HOST (n x 8 uint64, affine Montgomery; no device needed,
    ~0.3 ms per point): what a probe record's key points are compared with."""
    out = np.zeros((n, 8), dtype=np.uint64)
    lab = bytes(label)
    _lib.check(_lib.load().lurk_hip_ck_from_label_host(curve, lab, len(lab), n, _lib.ptr(out)))
    return out


def nifs_absorb_list(curve: int, pp_digest: int, comm_W1, comm_E1, u1_mont, x1_mont, comm_W2, x2_mont, comm_T) -> list[int]:
    """The elements ``nifs_challenge`` absorbs, in order (canonical integers of the RO's field) under the parameters in force."""
    a = lambda v: np.ascontiguousarray(v, dtype=np.uint64)
    dig = np.array([(pp_digest >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
    x1, x2 = a(x1_mont).reshape(-1, 4), a(x2_mont).reshape(-1, 4)
    assert len(x1) == len(x2)
    lib = _lib.load()
    cnt = ctypes.c_size_t()
    args = (curve, _lib.ptr(dig), _lib.ptr(a(comm_W1)), _lib.ptr(a(comm_E1)), _lib.ptr(a(u1_mont)), _lib.ptr(x1), _lib.ptr(a(comm_W2)), _lib.ptr(x2),
            len(x1), _lib.ptr(a(comm_T)))
    _lib.check(lib.lurk_hip_nifs_absorb_list(*args, None, 0, ctypes.byref(cnt)))
    out = np.zeros((cnt.value, 4), dtype=np.uint64)
    _lib.check(lib.lurk_hip_nifs_absorb_list(*args, _lib.ptr(out), cnt.value, ctypes.byref(cnt)))
    return [sum(int(r[k]) << (64 * k) for k in range(4)) for r in out]

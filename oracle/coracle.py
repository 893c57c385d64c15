"""ctypes binding of oracle/liblurk_oracle.so (the fast C oracle).  TEST INFRASTRUCTURE ONLY -
see the header of oracle/oracle.c.  Arrays are numpy uint64, 4 little-endian limbs per element."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import pyref

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblurk_oracle.so")
_lib = None

u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_on_curve_affine_mont.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def ints_to_limbs(vals) -> np.ndarray:
    """list of Python ints -> (n,4) uint64 limbs."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for w in range(4):
            out[i, w] = (int(v) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a: np.ndarray) -> list[int]:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a]


def to_mont(field: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_to_mont(field, _p(a), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def from_mont(field: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_from_mont(field, _p(a), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def mul_canonical(field: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Element-wise product of canonical field elements."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_mul_canonical(field, _p(a), _p(b), _p(out), ctypes.c_size_t(a.size // 4))
    return out


def dot(field: int, a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_dot_canonical(field, _p(a), _p(b), ctypes.c_size_t(a.size // 4), _p(out))
    return limbs_to_ints(out)[0]


def synth_scalars(field: int, stream: int, dist: int, n: int, first: int = 0) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_synth_scalars(field, ctypes.c_uint64(stream), dist, ctypes.c_size_t(first), ctypes.c_size_t(n), _p(out))
    return out


def synth_base_scalars(curve: int, n: int, first: int = 0) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_synth_base_scalars(curve, ctypes.c_size_t(first), ctypes.c_size_t(n), _p(out))
    return out


def fixed_base_mul(curve: int, k: np.ndarray) -> np.ndarray:
    """[k_i]G -> affine Montgomery (n, 8)."""
    k = np.ascontiguousarray(k, dtype=np.uint64)
    n = k.size // 4
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_fixed_base_mul(curve, _p(k), ctypes.c_size_t(n), _p(out))
    return out


def synth_bases(curve: int, n: int, first: int = 0) -> np.ndarray:
    return fixed_base_mul(curve, synth_base_scalars(curve, n, first))


_POS_CACHE = {}


def poseidon_consts(field: int, arity: int):
    key = (field, arity)
    if key not in _POS_CACHE:
        rf, rp = pyref.round_numbers(arity)
        rc = ints_to_limbs(pyref.round_constants(field, arity))
        mds = ints_to_limbs([x for row in pyref.mds_matrix(field, arity) for x in row])
        _POS_CACHE[key] = (rf, rp, rc, mds)
    return _POS_CACHE[key]


def poseidon_batch(field: int, arity: int, preimages: np.ndarray) -> np.ndarray:
    """preimages (n, arity, 4) canonical -> digests (n, 4) canonical."""
    pre = np.ascontiguousarray(preimages, dtype=np.uint64)
    n = pre.size // (4 * arity)
    rf, rp, rc, mds = poseidon_consts(field, arity)
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_poseidon_batch(field, arity, rf, rp, _p(rc), _p(mds), _p(pre), ctypes.c_size_t(n), _p(out))
    return out


def poseidon_tree8(field: int, leaves: np.ndarray, want_levels: bool = False):
    leaves = np.ascontiguousarray(leaves, dtype=np.uint64)
    n = leaves.size // 4
    rf, rp, rc, mds = poseidon_consts(field, 8)
    root = np.empty(4, dtype=np.uint64)
    total = 0
    m = n
    while m > 1:
        m //= 8
        total += m
    levels = np.empty((total, 4), dtype=np.uint64) if want_levels else None
    lib().orc_poseidon_tree8(field, rf, rp, _p(rc), _p(mds), _p(leaves), ctypes.c_size_t(n), _p(root),
                             _p(levels) if want_levels else None)
    return (root, levels) if want_levels else root


def store_hydrate(field: int, records: np.ndarray, values: np.ndarray):
    """StoreCore::hydrate_z_cache restated (oracle.c: orc_store_hydrate): records (n, 8) u32 {kind, tag, child[4], value, reserved},
    values (m, 4) u64 canonical -> (digests (n, 4) canonical, number of levels)."""
    rec = np.ascontiguousarray(records, dtype=np.uint32)
    vals = np.ascontiguousarray(values, dtype=np.uint64)
    n = rec.shape[0]
    consts = [poseidon_consts(field, a) for a in (3, 4, 6, 8)]
    rf = (ctypes.c_int * 4)(*[c[0] for c in consts])
    rp = (ctypes.c_int * 4)(*[c[1] for c in consts])
    rc = (ctypes.c_void_p * 4)(*[c[2].ctypes.data for c in consts])
    mds = (ctypes.c_void_p * 4)(*[c[3].ctypes.data for c in consts])
    out = np.empty((n, 4), dtype=np.uint64)
    levels = ctypes.c_size_t()
    lib().orc_store_hydrate(field, ctypes.c_void_p(rec.ctypes.data), ctypes.c_size_t(n), _p(vals), rf, rp, rc, mds, _p(out), ctypes.byref(levels))
    return out, levels.value


def msm_naive(curve: int, bases: np.ndarray, scalars: np.ndarray) -> np.ndarray:
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.empty(12, dtype=np.uint64)
    lib().orc_msm_naive(curve, _p(bases), _p(scalars), ctypes.c_size_t(scalars.size // 4), _p(out))
    return out


def msm_pippenger(curve: int, bases: np.ndarray, scalars: np.ndarray, nthreads: int = 0) -> np.ndarray:
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.empty(12, dtype=np.uint64)
    if nthreads <= 0:
        nthreads = lib().orc_num_threads()
    lib().orc_msm_pippenger(curve, _p(bases), _p(scalars), ctypes.c_size_t(scalars.size // 4), nthreads, _p(out))
    return out


def msm_fast(curve: int, bases: np.ndarray, scalars: np.ndarray, nthreads: int = 0, info: dict | None = None) -> np.ndarray:
    """The CPU baseline (oracle/msm_fast.c: pasta-msm-shaped Pippenger); same interface and result as msm_pippenger."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.empty(12, dtype=np.uint64)
    if nthreads <= 0:
        nthreads = lib().orc_num_threads()
    c, tiles = ctypes.c_int(), ctypes.c_int()
    lib().orc_msm_fast(curve, _p(bases), _p(scalars), ctypes.c_size_t(scalars.size // 4), nthreads, _p(out), ctypes.byref(c), ctypes.byref(tiles))
    if info is not None:
        info.update(window_bits=c.value, tiles=tiles.value, threads=nthreads)
    return out


def fast_mul_ns(curve: int = 0, iters: int = 2_000_000) -> float:
    lib().orc_fast_mul_ns.restype = ctypes.c_double
    return float(lib().orc_fast_mul_ns(curve, iters))


def jac_to_affine(curve: int, jac: np.ndarray) -> tuple[int, int]:
    """Jacobian Montgomery (12 limbs) -> canonical affine (x, y) ints; identity -> (0, 0)."""
    jac = np.ascontiguousarray(jac, dtype=np.uint64)
    out = np.empty(8, dtype=np.uint64)
    lib().orc_jac_mont_to_affine_canonical(curve, _p(jac), _p(out))
    x, y = limbs_to_ints(out)
    return x, y


def affine_to_ints(curve: int, aff_mont: np.ndarray) -> list[tuple[int, int]]:
    aff_mont = np.ascontiguousarray(aff_mont, dtype=np.uint64)
    n = aff_mont.size // 8
    out = np.empty_like(aff_mont)
    lib().orc_affine_mont_to_canonical(curve, _p(aff_mont), _p(out), ctypes.c_size_t(n))
    v = limbs_to_ints(out)
    return [(v[2 * i], v[2 * i + 1]) for i in range(n)]


def on_curve(curve: int, aff_mont: np.ndarray) -> bool:
    aff_mont = np.ascontiguousarray(aff_mont, dtype=np.uint64).reshape(-1, 8)
    return all(lib().orc_on_curve_affine_mont(curve, _p(np.ascontiguousarray(r))) for r in aff_mont)


def gen_mul(curve: int, k: int) -> np.ndarray:
    out = np.empty(12, dtype=np.uint64)
    lib().orc_gen_mul(curve, _p(ints_to_limbs([k])), _p(out))
    return out


def jac_add(curve: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.empty(12, dtype=np.uint64)
    lib().orc_jac_add(curve, _p(np.ascontiguousarray(a, dtype=np.uint64)), _p(np.ascontiguousarray(b, dtype=np.uint64)), _p(out))
    return out


def ntt(field: int, data: np.ndarray, inverse: bool = False) -> np.ndarray:
    a = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 4)
    log_n = int(a.shape[0]).bit_length() - 1
    assert 1 << log_n == a.shape[0]
    lib().orc_ntt(field, _p(a), log_n, int(inverse))
    return a


# ---- relaxed-R1CS folding ------------------------------------------------------
def spmv(field: int, indptr: np.ndarray, indices: np.ndarray, data: np.ndarray, z: np.ndarray) -> np.ndarray:
    """CSR (arecibo SparseMatrix {data, indices, indptr}) times z; everything canonical."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
    indices = np.ascontiguousarray(indices, dtype=np.uint64)
    data = np.ascontiguousarray(data, dtype=np.uint64)
    z = np.ascontiguousarray(z, dtype=np.uint64)
    rows = indptr.size - 1
    out = np.empty((rows, 4), dtype=np.uint64)
    lib().orc_spmv(field, ctypes.c_size_t(rows), _p(indptr), _p(indices), _p(data), _p(z), _p(out))
    return out


def cross_term(field: int, az1, bz1, cz1, az2, bz2, cz2, u1: int, u2: int) -> np.ndarray:
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (az1, bz1, cz1, az2, bz2, cz2)]
    rows = arrs[0].size // 4
    out = np.empty((rows, 4), dtype=np.uint64)
    lib().orc_cross_term(field, ctypes.c_size_t(rows), *[_p(a) for a in arrs], _p(ints_to_limbs([u1])), _p(ints_to_limbs([u2])), _p(out))
    return out


def axpy(field: int, a: np.ndarray, b: np.ndarray, r: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_axpy(field, ctypes.c_size_t(a.size // 4), _p(a), _p(b), _p(ints_to_limbs([r])), _p(out))
    return out


def relaxed_residual(field: int, az, bz, cz, u: int, e) -> np.ndarray:
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (az, bz, cz)]
    e = np.ascontiguousarray(e, dtype=np.uint64)
    out = np.empty_like(arrs[0])
    lib().orc_relaxed_residual(field, ctypes.c_size_t(arrs[0].size // 4), *[_p(a) for a in arrs], _p(ints_to_limbs([u])), _p(e), _p(out))
    return out


def _frame_structured_columns(rng, row_of_entry, num_cons, num_vars, num_io):
    """Column pattern of the Lurk step circuit: W = [globals | frame 0 aux | frame 1 aux | ...] (frames are
    synthesized independently and their aux concatenated, /root/reference/src/lem/multiframe.rs:699-702, 11 141
    constraints and 9 119 aux per frame, src/lem/eval.rs:1966-1967), so frame f's rows touch frame f's block (88 %),
    the globals at the front (6 %), the previous frame's block (4 %: its outputs) and the constant-one column u (2 %)."""
    import numpy as np

    nf = max(1, num_cons // 11141)
    cons_pf, vars_pf = -(-num_cons // nf), max(1, num_vars // nf)
    frame = np.minimum(row_of_entry // cons_pf, nf - 1)
    kind = rng.random(row_of_entry.size)
    local = frame * vars_pf + rng.integers(0, vars_pf, row_of_entry.size)
    prev = np.maximum(frame - 1, 0) * vars_pf + rng.integers(0, vars_pf, row_of_entry.size)
    glob = rng.integers(0, min(256, num_vars), row_of_entry.size)
    cols = np.where(kind < 0.88, local, np.where(kind < 0.94, glob, np.where(kind < 0.98, prev, num_vars)))
    return np.minimum(cols, num_vars + num_io).astype(np.uint64)


def synth_r1cs(field: int, num_cons: int, num_vars: int, num_io: int, seed: int = 7):
    """Synthetic R1CS shape of the kind the Lurk step circuit produces (3-4 entries per row, mostly +-1 and small
    coefficients, a few long rows) together with a strictly satisfying z2 = [W2 | 1 | X2]:
    A, B random sparse; C has one entry per row, on the u column, equal to (A z2)_i (B z2)_i.
    Returns (A, B, C, z2) with each matrix = (indptr u64, indices u64, data canonical (nnz, 4) u64)."""
    from . import pyref as R

    p = R.modulus(field)
    rng = np.random.default_rng(seed)
    ncols = num_vars + 1 + num_io
    table_ints = [1, p - 1, 2, p - 2, 3, 4, 8, 16, 256, 1 << 32, p - (1 << 16)] + [R.uniform_fe(90 + seed, i, p) for i in range(21)]
    table = ints_to_limbs(table_ints)
    weights = np.array([40, 25, 5, 2, 2, 1, 1, 1, 1, 1, 1] + [1] * 21, dtype=np.float64)
    weights /= weights.sum()

    def sparse():
        cnt = rng.integers(3, 5, num_cons).astype(np.uint64)
        long_rows = rng.integers(0, num_cons, max(1, num_cons // 300))  # bit-decomposition-like rows
        cnt[long_rows] = min(ncols, 255)
        indptr = np.zeros(num_cons + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        rows = np.repeat(np.arange(num_cons, dtype=np.int64), cnt.astype(np.int64))
        indices = _frame_structured_columns(rng, rows, num_cons, num_vars, num_io)
        data = table[rng.choice(len(table_ints), size=nnz, p=weights)]
        return indptr, indices, np.ascontiguousarray(data)

    A, B = sparse(), sparse()
    z2 = synth_scalars(field, 6, 0, ncols, first=seed * 1000003)
    z2[num_vars] = (1, 0, 0, 0)
    az, bz = spmv(field, *A, z2), spmv(field, *B, z2)
    cdata = np.empty((num_cons, 4), dtype=np.uint64)
    lib().orc_mul_canonical(field, _p(az), _p(bz), _p(cdata), ctypes.c_size_t(num_cons))
    C = (np.arange(num_cons + 1, dtype=np.uint64), np.full(num_cons, num_vars, dtype=np.uint64), cdata)
    return A, B, C, z2

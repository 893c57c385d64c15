"""Host-side mirror of one Nova folding step per curve over the HIP library.

Reference: ``RecursiveSNARK::prove_step`` as lurk-beta drives it (/root/reference/src/proof/nova.rs:282-295; SuperNova
/root/reference/src/proof/supernova.rs:231-244) = arecibo's ``NIFS::prove`` on the primary (Pallas: the Lurk step circuit)
and the secondary (Vesta) curve.  ``FoldingContext`` is one curve's half: the running relaxed pair stays in HBM, ``begin``
returns the two commitments the transcript absorbs, ``finish(r)`` folds.  The instance side (u, X and the two commitments of
the running instance) is folded here on the host exactly as ``RelaxedR1CSInstance::fold`` does: comm_W1 + r comm_W2,
comm_E1 + r comm_T, u1 + r, X1 + r X2 - the last two are read back from the device copy of z = [W | u | X].

``public_io`` is Z1: ``Store::to_scalar_vector`` (/root/reference/src/lem/store.rs:883-895), the step's input/output
[tag, hash] x (expr, env, cont)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .msm import point_sum


def public_io(z_ptrs) -> list[int]:
    """``to_scalar_vector``: z_ptrs = [(tag, hash), ...] -> [tag_0, hash_0, tag_1, hash_1, ...] (store.rs:883-895)."""
    out: list[int] = []
    for tag, h in z_ptrs:
        out += [int(tag), int(h)]
    return out


def point_mul(curve: int, point: np.ndarray, scalar: np.ndarray, is_mont: bool = True) -> np.ndarray:
    lib = _lib.load()
    out = np.zeros(12, dtype=np.uint64)
    p = np.ascontiguousarray(point, dtype=np.uint64)
    s = np.ascontiguousarray(scalar, dtype=np.uint64).reshape(4)
    _lib.check(lib.lurk_hip_point_mul(curve, _lib.ptr(out), _lib.ptr(p), _lib.ptr(s), int(is_mont)))
    return out


class FoldingContext:
    """One curve of the cycle: R1CS shape + commitment key (both resident, borrowed) + the running relaxed pair."""

    def __init__(self, curve: int, shape, key):
        lib = _lib.load()
        self.curve, self.shape, self.key = curve, shape, key
        self._h = ctypes.c_void_p()
        _lib.check(lib.lurk_hip_fold_ctx_create(ctypes.byref(self._h), curve, shape._h, key._ctx))
        ident = np.zeros(12, dtype=np.uint64)
        self.comm_W, self.comm_E = ident.copy(), ident.copy()  # RelaxedR1CSInstance::default: identity commitments

    def set_running(self, z1: np.ndarray, e1: np.ndarray, comm_W: np.ndarray, comm_E: np.ndarray):
        z1 = np.ascontiguousarray(z1, dtype=np.uint64)
        e1 = np.ascontiguousarray(e1, dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_fold_ctx_set_running(self._h, _lib.ptr(z1), _lib.ptr(e1)))
        self.comm_W, self.comm_E = np.array(comm_W, dtype=np.uint64), np.array(comm_E, dtype=np.uint64)

    def begin(self, w2, x2_mont: np.ndarray, stream=None):
        """w2: host (num_vars, 4) u64 array or a device tensor (Montgomery).  Returns (comm_W2, comm_T), 96-byte Jacobians."""
        lib = _lib.load()
        on_dev = hasattr(w2, "data_ptr")
        if not on_dev:
            w2 = np.ascontiguousarray(w2, dtype=np.uint64)
        x2 = np.ascontiguousarray(x2_mont, dtype=np.uint64)
        cw, ct = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_fold_step_begin(self._h, _lib.ptr(w2), int(on_dev), _lib.ptr(stream), _lib.ptr(x2), _lib.ptr(cw), _lib.ptr(ct)))
        self._open = (cw, ct)
        return cw, ct

    def prefetch(self, w2_range, offset: int = 0, stream=None):
        """Stage positions [offset, offset + len) of the NEXT fresh witness (host array or device tensor, Montgomery; the rest
        zero until ``begin_prefetched`` supplies it) and start its commitment under whatever the device is doing now."""
        on_dev = hasattr(w2_range, "data_ptr")
        if on_dev:
            count = w2_range.numel() // 4
        else:
            w2_range = np.ascontiguousarray(w2_range, dtype=np.uint64)
            count = w2_range.size // 4
        _lib.check(_lib.load().lurk_hip_fold_step_prefetch(self._h, _lib.ptr(w2_range), offset, count, int(on_dev), _lib.ptr(stream)))

    def begin_prefetched(self, x2_mont: np.ndarray, patches=()):
        """Open the step of the oldest staged instance.  patches: [(offset, host (k, 4) u64 array)], the ranges of W2 known only
        now.  Returns (comm_W2, comm_T)."""

        class Patch(ctypes.Structure):
            _fields_ = [("offset", ctypes.c_size_t), ("count", ctypes.c_size_t), ("values", ctypes.c_void_p)]

        keep = [np.ascontiguousarray(v, dtype=np.uint64) for _, v in patches]
        arr = (Patch * max(1, len(keep)))()
        for k, ((off, _), v) in enumerate(zip(patches, keep)):
            arr[k] = Patch(int(off), v.size // 4, v.ctypes.data)
        x2 = np.ascontiguousarray(x2_mont, dtype=np.uint64)
        cw, ct = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_fold_step_begin_prefetched(self._h, ctypes.cast(arr, ctypes.c_void_p), len(keep), _lib.ptr(x2),
                                                                    _lib.ptr(cw), _lib.ptr(ct)))
        self._open = (cw, ct)
        return cw, ct

    def finish(self, r_mont: np.ndarray):
        """Folds the witness pair on the device and the running instance's commitments on the host."""
        r = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
        _lib.check(_lib.load().lurk_hip_fold_step_finish(self._h, _lib.ptr(r)))
        cw, ct = self._open
        self.comm_W = point_sum(self.curve, np.stack([self.comm_W, point_mul(self.curve, cw, r)]))
        self.comm_E = point_sum(self.curve, np.stack([self.comm_E, point_mul(self.curve, ct, r)]))

    def running_device(self):
        """(z pointer, E pointer, stream) of the running pair in HBM (valid until the next finish)."""
        z, e, s = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(_lib.load().lurk_hip_fold_ctx_running_dev(self._h, ctypes.byref(z), ctypes.byref(e), ctypes.byref(s)))
        return z.value, e.value, s.value

    def read(self):
        """Host copies (z = [W | u | X], E), Montgomery."""
        z = np.zeros((self.shape.num_cols, 4), dtype=np.uint64)
        e = np.zeros((self.shape.num_cons, 4), dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_fold_ctx_read(self._h, _lib.ptr(z), _lib.ptr(e)))
        return z, e

    def close(self):
        if self._h:
            _lib.check(_lib.load().lurk_hip_fold_ctx_destroy(self._h))
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

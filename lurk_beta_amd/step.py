"""Host-side mirror of one Nova folding step per curve over the HIP library.

Reference: ``RecursiveSNARK::prove_step`` as lurk-beta drives it (/root/reference/src/proof/nova.rs:282-295; SuperNova
/root/reference/src/proof/supernova.rs:231-244) = arecibo's ``NIFS::prove`` on the primary (Pallas: the Lurk step circuit)
and the secondary (Vesta) curve.  ``FoldingContext`` is one curve's half: the running relaxed pair stays in HBM, ``begin``
returns the two commitments the transcript absorbs, ``finish(r)`` folds; ``step`` is the whole of NIFS::prove with the
challenge derived by the library's own transcript (arecibo's PoseidonRO, ``lurk_hip_nifs_challenge``).  The instance side (u, X
and the two commitments of the running instance) is folded inside the library, on the host, exactly as
``RelaxedR1CSInstance::fold`` does: comm_W1 + r comm_W2, comm_E1 + r comm_T, u1 + r, X1 + r X2.
``NivcFoldingContext`` is SuperNova's arrangement (/root/reference/src/proof/supernova.rs:226-244): one running pair per
circuit index over ONE commitment key, the step's ``pc`` picks the pair that folds.

``public_io`` is Z1: ``Store::to_scalar_vector`` (/root/reference/src/lem/store.rs:883-895), the step's input/output
[tag, hash] x (expr, env, cont)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


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


class _W2Patch(ctypes.Structure):
    """lurk_hip_w2_patch (module level: building a ctypes Structure class per call cost the step ~50 us of host time on its serial chain)"""
    _fields_ = [("offset", ctypes.c_size_t), ("count", ctypes.c_size_t), ("values", ctypes.c_void_p)]


class FoldingContext:
    """One curve of the cycle: R1CS shape + commitment key (both resident, borrowed) + the running relaxed pair."""

    def __init__(self, curve: int, shape, key):
        lib = _lib.load()
        self.curve, self.shape, self.key = curve, shape, key
        self._h = ctypes.c_void_p()
        from .msm import MultiCommitmentKey

        if isinstance(key, MultiCommitmentKey):  # the key cut across several devices: slices commit concurrently, partials summed on the host
            _lib.check(lib.lurk_hip_fold_ctx_create_multi(ctypes.byref(self._h), curve, shape._h, key._ctx))
        else:
            _lib.check(lib.lurk_hip_fold_ctx_create(ctypes.byref(self._h), curve, shape._h, key._ctx))

    def instance(self):
        """The running instance U = (comm_W, comm_E, u, X): 96-byte Jacobians, Montgomery scalars."""
        cw, ce = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        u, x = np.zeros(4, dtype=np.uint64), np.zeros((max(self.shape.num_io, 1), 4), dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_fold_ctx_instance(self._h, _lib.ptr(cw), _lib.ptr(ce), _lib.ptr(u), _lib.ptr(x)))
        return cw, ce, u, x[: self.shape.num_io]

    @property
    def comm_W(self):
        return self.instance()[0]

    @property
    def comm_E(self):
        return self.instance()[1]

    def set_running(self, z1: np.ndarray, e1: np.ndarray, comm_W: np.ndarray, comm_E: np.ndarray):
        z1 = np.ascontiguousarray(z1, dtype=np.uint64)
        e1 = np.ascontiguousarray(e1, dtype=np.uint64)
        cw, ce = np.ascontiguousarray(comm_W, dtype=np.uint64), np.ascontiguousarray(comm_E, dtype=np.uint64)
        lib = _lib.load()
        _lib.check(lib.lurk_hip_fold_ctx_set_running(self._h, _lib.ptr(z1), _lib.ptr(e1)))
        _lib.check(lib.lurk_hip_fold_ctx_set_instance(self._h, _lib.ptr(cw), _lib.ptr(ce)))

    def step(self, w2, x2_mont: np.ndarray, pp_digest: int, stream=None):
        """NIFS::prove whole: both commitments, r = RO(pp_digest, U1, U2, comm_T) from the library's transcript, the fold.
        Returns (comm_W2, comm_T, r) with r in Montgomery form."""
        on_dev = hasattr(w2, "data_ptr")
        if not on_dev:
            w2 = np.ascontiguousarray(w2, dtype=np.uint64)
        x2 = np.ascontiguousarray(x2_mont, dtype=np.uint64)
        dig = np.array([(pp_digest >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
        cw, ct, r = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        self._check_begin(_lib.load().lurk_hip_fold_step(self._h, _lib.ptr(w2), int(on_dev), _lib.ptr(stream), _lib.ptr(x2), _lib.ptr(dig), _lib.ptr(cw),
                                                  _lib.ptr(ct), _lib.ptr(r)))
        return cw, ct, r

    def begin(self, w2, x2_mont: np.ndarray, stream=None):
        """w2: host (num_vars, 4) u64 array or a device tensor (Montgomery).  Returns (comm_W2, comm_T), 96-byte Jacobians."""
        lib = _lib.load()
        on_dev = hasattr(w2, "data_ptr")
        if not on_dev:
            w2 = np.ascontiguousarray(w2, dtype=np.uint64)
        x2 = np.ascontiguousarray(x2_mont, dtype=np.uint64)
        cw, ct = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        self._check_begin(lib.lurk_hip_fold_step_begin(self._h, _lib.ptr(w2), int(on_dev), _lib.ptr(stream), _lib.ptr(x2), _lib.ptr(cw), _lib.ptr(ct)))
        self._open = (cw, ct)
        return cw, ct

    def set_pp_digest(self, pp_digest: int):
        """The digest of the public parameters, once: every ``begin`` then stages the transcript (U1 while the device works, U2 when
        comm_W2 is there) and ``challenge()`` finishes it behind comm_T with one permutation."""
        dig = np.array([(pp_digest >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_fold_ctx_set_pp_digest(self._h, _lib.ptr(dig)))

    def challenge(self) -> np.ndarray:
        """r = RO(pp_digest, U1, U2, comm_T) of the open step (Montgomery, ready for ``finish``) = ``nifs_challenge`` of the same values."""
        r = np.zeros(4, dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_fold_step_challenge(self._h, _lib.ptr(r)))
        return r

    def _check_begin(self, rc):
        err = getattr(self, "_hook_error", None)
        if err is not None:  # the submit hook raised: the library has rolled the step back (rc != 0); hand the caller its own exception
            self._hook_error = None
            raise err
        _lib.check(rc)

    def set_submit_hook(self, fn):
        """``fn()`` is called once per step from inside ``begin`` / ``begin_prefetched`` / ``step``, after the step's device work has been
        enqueued and before the call blocks on the commitments: the place to enqueue the next witness's slot traces (they queue behind the
        step's opening kernels and fill what its commitments leave).  ``None`` removes the hook.  An exception raised by ``fn`` fails
        the begin (the step is rolled back) and is re-raised."""
        import ctypes

        if fn is None:
            _lib.check(_lib.load().lurk_hip_fold_ctx_set_submit_hook(self._h, None, None))
            self._hook = None
            return
        self._hook_error = None

        def tramp(_user):
            try:
                fn()
                return 0
            except BaseException as e:  # noqa: BLE001 - an exception must not unwind through the C frames
                self._hook_error = e
                return 1

        cb = _lib.FOLD_SUBMIT_HOOK_FN(tramp)
        _lib.check(_lib.load().lurk_hip_fold_ctx_set_submit_hook(self._h, ctypes.cast(cb, ctypes.c_void_p), None))
        self._hook = cb  # (the library keeps the pointer: the trampoline lives as long as the context)

    def add_helper(self, helper_key):
        """Staging ahead across devices: ``helper_key`` is a ``CommitmentKey`` over the same bases resident on another device; instances
        staged with ``prefetch`` are committed on the helpers in turn (peer copy of the staged ranges) while this context's device folds."""
        _lib.check(_lib.load().lurk_hip_fold_ctx_add_helper(self._h, helper_key._ctx))
        self._helpers = getattr(self, "_helpers", []) + [helper_key]

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
        keep = [np.ascontiguousarray(v, dtype=np.uint64) for _, v in patches]
        arr = (_W2Patch * max(1, len(keep)))()
        for k, ((off, _), v) in enumerate(zip(patches, keep)):
            arr[k] = _W2Patch(int(off), v.size // 4, v.ctypes.data)
        x2 = np.ascontiguousarray(x2_mont, dtype=np.uint64)
        cw, ct = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        self._check_begin(_lib.load().lurk_hip_fold_step_begin_prefetched(self._h, ctypes.cast(arr, ctypes.c_void_p), len(keep), _lib.ptr(x2),
                                                                    _lib.ptr(cw), _lib.ptr(ct)))
        self._open = (cw, ct)
        return cw, ct

    def finish(self, r_mont: np.ndarray):
        """Folds the witness pair on the device and the running instance (commitments, u, X) on the host, inside the library."""
        r = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
        _lib.check(_lib.load().lurk_hip_fold_step_finish(self._h, _lib.ptr(r)))

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


def nifs_challenge(curve: int, pp_digest: int, comm_W1, comm_E1, u1_mont, x1_mont, comm_W2, x2_mont, comm_T) -> np.ndarray:
    """r = RO(pp_digest, U1, U2, comm_T) as NIFS::prove derives it (host code of the library; Montgomery form out)."""
    a = lambda v: np.ascontiguousarray(v, dtype=np.uint64)
    dig = np.array([(pp_digest >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
    x1, x2 = a(x1_mont).reshape(-1, 4), a(x2_mont).reshape(-1, 4)
    assert len(x1) == len(x2)
    r = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.load().lurk_hip_nifs_challenge(curve, _lib.ptr(dig), _lib.ptr(a(comm_W1)), _lib.ptr(a(comm_E1)), _lib.ptr(a(u1_mont)),
                                                   _lib.ptr(x1), _lib.ptr(a(comm_W2)), _lib.ptr(x2), len(x1), _lib.ptr(a(comm_T)), _lib.ptr(r)))
    return r


def nova_ro_squeeze(field_id: int, elems: list[int], num_bits: int = 128) -> int:
    """arecibo ``PoseidonRO``: absorb canonical elements of ``field_id``, squeeze ``num_bits`` bits."""
    arr = np.array([[(e >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for e in elems], dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.load().lurk_hip_nova_ro_squeeze(field_id, _lib.ptr(arr), len(elems), num_bits, _lib.ptr(out)))
    return sum(int(out[k]) << (64 * k) for k in range(4))


class NivcFoldingContext:
    """SuperNova's non-uniform IVC on one curve (/root/reference/src/proof/supernova.rs:226-244; circuit selection
    /root/reference/src/lem/multiframe.rs:271-356): one R1CS shape and one running relaxed pair PER circuit index, all
    under ONE commitment key (sized for the largest circuit); a step names its circuit with ``pc`` and folds into that
    circuit's running pair only, the others stay as they are.  ``shapes[i]`` may be None for a circuit that is never run."""

    def __init__(self, curve: int, shapes, key):
        self.curve, self.key = curve, key
        self.ctxs = [FoldingContext(curve, sh, key) if sh is not None else None for sh in shapes]
        self.pc_trace: list[int] = []

    def step(self, pc: int, w2, x2_mont, pp_digest: int, stream=None):
        if not 0 <= pc < len(self.ctxs) or self.ctxs[pc] is None:
            raise ValueError(f"no circuit with index {pc}")
        self.pc_trace.append(pc)
        return self.ctxs[pc].step(w2, x2_mont, pp_digest, stream=stream)

    def __getitem__(self, pc: int) -> FoldingContext:
        return self.ctxs[pc]

    def close(self):
        for c in self.ctxs:
            if c is not None:
                c.close()

"""Host-side mirror of the Pedersen commitment path over the HIP library.

Mirrors arecibo's ``CommitmentKey`` / ``CommitmentEngine::commit(ck, v)`` =
``vartime_multiscalar_mul(v, &ck[..v.len()])`` as lurk-beta reaches it through
``RecursiveSNARK::new`` / ``prove_step`` (/root/reference/src/proof/nova.rs:287-293,
/root/reference/src/proof/supernova.rs:231-244).  The commitment key is uploaded once and stays
resident in HBM (``lurk_hip_msm_ctx_*``); ``commit`` streams only the scalars."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

CURVE_SCALAR_FIELD = {0: 1, 1: 0}  # Pallas scalars live in Fq, Vesta scalars in Fp
CURVE_BASE_FIELD = {0: 0, 1: 1}


def msm(curve: int, bases: np.ndarray, scalars: np.ndarray, is_mont: bool = False) -> np.ndarray:
    """One-shot ``mult_pippenger_{pallas,vesta}``: host buffers in, 96-byte Jacobian out (12 u64)."""
    lib = _lib.load()
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.size // 4
    assert bases.size // 8 == n, "bases and scalars differ in length"
    out = np.zeros(12, dtype=np.uint64)
    fn = lib.lurk_hip_msm_pallas if curve == 0 else lib.lurk_hip_msm_vesta
    _lib.check(fn(_lib.ptr(out), _lib.ptr(bases), n, _lib.ptr(scalars), int(is_mont)))
    return out


def point_to_affine(curve: int, jac: np.ndarray) -> tuple[int, int]:
    """Jacobian (Montgomery) -> canonical affine (x, y) ints; identity -> (0, 0)."""
    lib = _lib.load()
    jac = np.ascontiguousarray(jac, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    _lib.check(lib.lurk_hip_point_to_affine_canonical(curve, _lib.ptr(out), _lib.ptr(jac)))
    v = [int(out[4 * k]) | int(out[4 * k + 1]) << 64 | int(out[4 * k + 2]) << 128 | int(out[4 * k + 3]) << 192 for k in range(2)]
    return v[0], v[1]


def point_sum(curve: int, points: np.ndarray) -> np.ndarray:
    """Sum of Jacobian points (count x 12 u64) -> Jacobian.  Used to fold per-rank partial commitments."""
    lib = _lib.load()
    points = np.ascontiguousarray(points, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(lib.lurk_hip_point_sum(curve, _lib.ptr(out), _lib.ptr(points), points.size // 12))
    return out


def point_sum_gathered(curve: int, gathered: np.ndarray) -> np.ndarray:
    """``lurk_hip_point_sum_gathered``: the commitment of a vector from the world x 12 u64 partial commitments of its slices (rank order) -
    the one library call a one-process-per-GPU host needs after its own all-gather."""
    gathered = np.ascontiguousarray(gathered, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(_lib.load().lurk_hip_point_sum_gathered(curve, _lib.ptr(out), _lib.ptr(gathered), gathered.size // 12))
    return out


class CommitmentKey:
    """Resident commitment key (``ck``): n affine bases kept in HBM for the lifetime of the object."""

    def __init__(self, curve: int, bases, n: int | None = None, precompute: bool = False, device: bool = False, stream=None,
                 window_bits: int = 0, small_form: bool | None = None):
        """small_form: None = the library decides for keys of <= 2^16 points (by the memory that is free); True = the small-commitment
        form or an error (LURK_MSM_FLAG_SMALL_FORM); False = never (LURK_MSM_FLAG_NO_SMALL_FORM)"""
        lib = _lib.load()
        self.curve = curve
        self._ctx = ctypes.c_void_p()
        flags = (1 if precompute else 0) | ((window_bits & 0xFF) << 8) | ((1 << 16) if small_form else (1 << 17) if small_form is False else 0)
        if device:
            assert n is not None
            self.n = n
            self._keepalive = bases  # borrowed device memory must outlive the ctx
            _lib.check(lib.lurk_hip_msm_ctx_create_dev(ctypes.byref(self._ctx), curve, _lib.ptr(bases), n, flags, _lib.ptr(stream)))
        else:
            bases = np.ascontiguousarray(bases, dtype=np.uint64)
            self.n = bases.size // 8
            _lib.check(lib.lurk_hip_msm_ctx_create(ctypes.byref(self._ctx), curve, _lib.ptr(bases), self.n, flags))

    @classmethod
    def load(cls, path: str, precompute: bool = False, window_bits: int = 0) -> "CommitmentKey":
        """Key file -> resident context (``lurk_hip_msm_ctx_load``): the public-parameter cache of the commitment path
        (/root/reference/src/public_parameters/mod.rs:33-56)."""
        lib = _lib.load()
        self = cls.__new__(cls)
        self._ctx = ctypes.c_void_p()
        flags = (1 if precompute else 0) | ((window_bits & 0xFF) << 8)
        _lib.check(lib.lurk_hip_msm_ctx_load(ctypes.byref(self._ctx), path.encode(), flags))
        info = self.info()
        self.curve, self.n = info["curve"], info["npoints"]
        return self

    @classmethod
    def from_label(cls, curve: int, label: bytes, n: int, precompute: bool = False, window_bits: int = 0) -> "CommitmentKey":
        """``CommitmentEngine::setup(label, n)`` (arecibo ``from_label``; /root/reference/src/proof/nova.rs:196-216) generated on the
        device straight into the resident context."""
        lib = _lib.load()
        self = cls.__new__(cls)
        self._ctx = ctypes.c_void_p()
        flags = (1 if precompute else 0) | ((window_bits & 0xFF) << 8)
        lab = bytes(label)
        _lib.check(lib.lurk_hip_msm_ctx_from_label(ctypes.byref(self._ctx), curve, lab, len(lab), n, flags))
        self.curve, self.n = curve, n
        return self

    def reserve(self, n: int, slots: int = 3) -> None:
        """Allocate the workspaces of slots 0..slots-1 for n-scalar commitments now instead of on first use."""
        _lib.check(_lib.load().lurk_hip_msm_ctx_reserve(self._ctx, n, slots))

    def save(self, path: str, with_table: bool = False) -> None:
        _lib.check(_lib.load().lurk_hip_msm_ctx_save(self._ctx, path.encode(), int(with_table)))

    def info(self) -> dict:
        c, n, w, p = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int()
        f = ctypes.c_int()
        _lib.check(_lib.load().lurk_hip_msm_ctx_info(self._ctx, ctypes.byref(c), ctypes.byref(n), ctypes.byref(w), ctypes.byref(p)))
        _lib.check(_lib.load().lurk_hip_msm_ctx_form(self._ctx, ctypes.byref(f)))
        return {"curve": c.value, "npoints": n.value, "window_bits": w.value, "precomputed": bool(p.value),
                "form": {0: "plain", 1: "table", 2: "small"}[f.value]}

    def commit(self, scalars: np.ndarray, is_mont: bool = False) -> np.ndarray:
        """``CE::commit(ck, v)``: host scalars (len <= n) -> Jacobian commitment."""
        lib = _lib.load()
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_msm_ctx_run(self._ctx, _lib.ptr(out), _lib.ptr(scalars), scalars.size // 4, int(is_mont)))
        return out

    def commit_device(self, d_scalars, n: int, is_mont: bool = False, stream=None) -> np.ndarray:
        """Scalars already resident in HBM (a torch tensor or a raw device pointer)."""
        lib = _lib.load()
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_msm_ctx_run_dev(self._ctx, _lib.ptr(out), _lib.ptr(d_scalars), n, int(is_mont), _lib.ptr(stream)))
        return out

    def submit_device(self, slot: int, d_scalars, n: int, is_mont: bool = False, stream=None, mode: int = 0) -> None:
        """Asynchronous commit on `slot` (0..3); pair with ``wait(slot)``.  mode: 0 = default, 1 = foreground (the commitment the
        host waits for next), 2 = background (work staged ahead: persistent one-wave accumulation beside the foreground one), 3 = follow
        (work staged ahead: low-priority sort and plan at once, accumulation once the latest foreground commitment's has ended)."""
        lib = _lib.load()
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = d_scalars  # the scalars must stay alive until wait()
        _lib.check(lib.lurk_hip_msm_ctx_submit_dev_mode(self._ctx, slot, _lib.ptr(d_scalars), n, int(is_mont), _lib.ptr(stream), mode))

    def wait(self, slot: int) -> np.ndarray:
        lib = _lib.load()
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_msm_ctx_wait(self._ctx, slot, _lib.ptr(out)))
        getattr(self, "_keep", {}).pop(slot, None)
        return out

    def fold_key(self, n: int, weights_mont: np.ndarray, stream=None):
        """The key folded by the weights of k inner-product rounds at once (lurk_hip_msm_ctx_fold_key_dev): a device tensor of
        m = n / len(weights) affine Montgomery points, out[p] = sum_b weights[b] * key[b m + p].  Window-table form only."""
        import torch

        w = np.ascontiguousarray(weights_mont, dtype=np.uint64).reshape(-1, 4)
        out = torch.empty((n // w.shape[0], 8), dtype=torch.int64, device="cuda")
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().lurk_hip_msm_ctx_fold_key_dev(self._ctx, n, _lib.ptr(w), w.shape[0], _lib.ptr(out), _lib.ptr(s)))
        return out

    def supports_pairs(self) -> bool:
        """True for a window-table key (precompute flag and more than 2^16 points, or a window-bit override): the only form that
        commits a pair in one pass (``submit_pair_device``)."""
        return self.info()["form"] == "table"

    def submit_pair_device(self, slot: int, d_scalars, n: int, sel_bit: int, is_mont: bool = False, stream=None) -> None:
        """Two commitments with disjoint supports in one pass: scalars whose index has bit ``sel_bit`` clear / set; ``wait_pair``."""
        lib = _lib.load()
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = d_scalars
        _lib.check(lib.lurk_hip_msm_ctx_submit_pair_dev(self._ctx, slot, _lib.ptr(d_scalars), n, int(is_mont), _lib.ptr(stream), sel_bit))

    def wait_pair(self, slot: int):
        """(commitment of the bit-clear scalars, commitment of the bit-set scalars)"""
        lib = _lib.load()
        lo, hi = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_msm_ctx_wait_pair(self._ctx, slot, _lib.ptr(lo), _lib.ptr(hi)))
        getattr(self, "_keep", {}).pop(slot, None)
        return lo, hi

    def close(self):
        if self._ctx:
            _lib.load().lurk_hip_msm_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiCommitmentKey:
    """``lurk_hip_msm_multi_*``: one process, a list of devices.  The key is cut into ``len(devices)`` contiguous
    slices (slice i resident on ``devices[i]``); ``commit`` runs the slices concurrently inside the library (one host
    thread per device) and sums the 96-byte partial commitments on the host.  This is the form a single-process
    prover binds (/root/reference/src/proof/nova.rs:304-326); ``distributed.ShardedCommitmentKey`` is the
    one-process-per-GPU form of the same sharding."""

    def __init__(self, curve: int, bases: np.ndarray, devices, precompute: bool = False, window_bits: int = 0, auto_slices: bool = False):
        lib = _lib.load()
        self.curve = curve
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        self.n = bases.size // 8
        self.devices = list(devices)
        devs = (ctypes.c_int * len(self.devices))(*self.devices)
        flags = (1 if precompute else 0) | ((window_bits & 0xFF) << 8) | ((1 << 18) if auto_slices else 0)  # LURK_MSM_FLAG_AUTO_SLICES
        self._ctx = ctypes.c_void_p()
        _lib.check(lib.lurk_hip_msm_multi_create(ctypes.byref(self._ctx), curve, _lib.ptr(bases), self.n, devs, len(self.devices), flags))

    def shards(self) -> list[tuple[int, int, int]]:
        """(device, first point, count) per slice."""
        lib = _lib.load()
        out = []
        for i in range(lib.lurk_hip_msm_multi_num_shards(self._ctx)):
            d, f, c = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(lib.lurk_hip_msm_multi_shard(self._ctx, i, ctypes.byref(d), ctypes.byref(f), ctypes.byref(c)))
            out.append((d.value, f.value, c.value))
        return out

    def commit(self, scalars: np.ndarray, is_mont: bool = False) -> np.ndarray:
        lib = _lib.load()
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_msm_multi_commit(self._ctx, _lib.ptr(out), _lib.ptr(scalars), scalars.size // 4, int(is_mont)))
        return out

    def commit_device(self, d_slices, n: int, is_mont: bool = False) -> np.ndarray:
        """``d_slices[i]``: slice i's scalars resident on slice i's device (torch tensors or raw pointers)."""
        lib = _lib.load()
        ptrs = (ctypes.c_void_p * len(d_slices))(*[_lib.ptr(x) for x in d_slices])
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.lurk_hip_msm_multi_commit_dev(self._ctx, _lib.ptr(out), ptrs, len(d_slices), n, int(is_mont)))
        return out

    def submit_device(self, slot: int, d_slices, n: int, is_mont: bool = False, streams=None, mode: int = 0) -> None:
        """Asynchronous form: slice i's scalars on slice i's device, ordered after ``streams[i]`` (a stream of that device)."""
        lib = _lib.load()
        ptrs = (ctypes.c_void_p * len(d_slices))(*[_lib.ptr(x) for x in d_slices])
        strs = (ctypes.c_void_p * len(d_slices))(*[_lib.ptr(x) for x in (streams or [None] * len(d_slices))])
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = d_slices
        _lib.check(lib.lurk_hip_msm_multi_submit_dev(self._ctx, slot, ptrs, strs, len(d_slices), n, int(is_mont), mode))

    def wait(self, slot: int) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(_lib.load().lurk_hip_msm_multi_wait(self._ctx, slot, _lib.ptr(out)))
        getattr(self, "_keep", {}).pop(slot, None)
        return out

    def close(self):
        if self._ctx:
            _lib.load().lurk_hip_msm_multi_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shake256(data: bytes, out_len: int) -> bytes:
    """The library's host-side SHAKE256 (the XOF behind ``from_label``); no GPU needed."""
    out = ctypes.create_string_buffer(out_len)
    _lib.check(_lib.load().lurk_hip_shake256(data, len(data), out, out_len))
    return out.raw


def ck_from_label(curve: int, label: bytes, n: int):
    """``from_label`` into a fresh device tensor of n affine Montgomery points (n, 8) int64."""
    import torch

    out = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    lab = bytes(label)
    _lib.check(_lib.load().lurk_hip_ck_from_label_dev(curve, lab, len(lab), n, _lib.ptr(out), _lib.ptr(torch.cuda.current_stream().cuda_stream)))
    return out

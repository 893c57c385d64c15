"""Device-resident relaxed-R1CS folding: the host-side mirror of the three arecibo
operations a Nova folding step runs between its two commitments -
``R1CSShape::multiply_vec``, the cross term of ``R1CSShape::commit_T`` and ``RelaxedR1CSWitness::fold`` -
as called from ``RecursiveSNARK::prove_step`` (/root/reference/src/proof/nova.rs:291-293; arecibo is the
un-vendored ``nova`` dependency of /root/reference/Cargo.toml:128).  All arithmetic runs in liblurk_hip.so;
vectors are torch device tensors of shape (n, 4) int64 holding 32-byte Montgomery field elements."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


class R1CSShape:
    """CSR matrices A, B, C (each ``(indptr, indices, data)`` as arecibo's SparseMatrix: usize row pointers,
    usize column indices into z = [W | u | X], 32-byte Montgomery values) kept resident on the GPU."""

    def __init__(self, field_id: int, num_cons: int, num_vars: int, num_io: int, A, B, C):
        lib = _lib.load()
        self.field_id, self.num_cons, self.num_vars, self.num_io = field_id, num_cons, num_vars, num_io
        args = []
        self._keep = []
        for indptr, indices, data in (A, B, C):
            ip = np.ascontiguousarray(indptr, dtype=np.uint64)
            ix = np.ascontiguousarray(indices, dtype=np.uint64)
            dv = np.ascontiguousarray(data, dtype=np.uint64)
            if ip.size != num_cons + 1 or ix.size * 4 != dv.size:
                raise ValueError("malformed CSR matrix")
            self._keep += [ip, ix, dv]
            args += [_lib.ptr(ip), _lib.ptr(ix), _lib.ptr(dv)]
        self._h = ctypes.c_void_p()
        _lib.check(lib.lurk_hip_r1cs_create(ctypes.byref(self._h), field_id, num_cons, num_vars, num_io, *args))
        self._keep = None  # the library copied everything

    @property
    def num_cols(self) -> int:
        return self.num_vars + 1 + self.num_io

    def info(self) -> dict:
        v = [ctypes.c_size_t() for _ in range(4)]
        _lib.check(_lib.load().lurk_hip_r1cs_info(self._h, *[ctypes.byref(x) for x in v]))
        return {"nnz": (v[0].value, v[1].value, v[2].value), "distinct_coefficients": v[3].value}

    def multiply_vec(self, d_z, stream=None):
        """(A z, B z, C z) - R1CSShape::multiply_vec."""
        import torch

        assert d_z.is_cuda and d_z.shape[0] == self.num_cols
        out = [torch.empty((self.num_cons, 4), dtype=torch.int64, device=d_z.device) for _ in range(3)]
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().lurk_hip_r1cs_multiply_vec_dev(self._h, _lib.ptr(d_z), *[_lib.ptr(o) for o in out], _lib.ptr(s)))
        return out

    def cross_term(self, d_z1, d_z2, out=None, stream=None):
        """T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1 - the vector R1CSShape::commit_T commits to."""
        import torch

        assert d_z1.is_cuda and d_z2.is_cuda and d_z1.shape[0] == self.num_cols and d_z2.shape[0] == self.num_cols
        if out is None:
            out = torch.empty((self.num_cons, 4), dtype=torch.int64, device=d_z1.device)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().lurk_hip_r1cs_cross_term_dev(self._h, _lib.ptr(d_z1), _lib.ptr(d_z2), _lib.ptr(out), _lib.ptr(s)))
        return out

    def cross_term_cached(self, d_z2, d_abc1, u1_mont, prev=None, r_prev_mont=None, stream=None):
        """The same T from the running instance's cached products (A z1, B z1, C z1) and z2 alone; u1_mont: the running u (4 x u64, host).
        prev = [A z2, B z2, C z2] of the previous step with its challenge r_prev_mont: folded into the cache IN PLACE by the same launch.
        Returns (T, [A z2, B z2, C z2]) - ``lurk_hip_r1cs_cross_term_cached_dev``."""
        import torch

        assert d_z2.is_cuda and d_z2.shape[0] == self.num_cols and len(d_abc1) == 3
        t = torch.empty((self.num_cons, 4), dtype=torch.int64, device=d_z2.device)
        abc2 = [torch.empty_like(t) for _ in range(3)]
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        u1 = np.ascontiguousarray(u1_mont, dtype=np.uint64).reshape(4)
        rp = None if r_prev_mont is None else np.ascontiguousarray(r_prev_mont, dtype=np.uint64).reshape(4)
        pv = [None, None, None] if prev is None else list(prev)
        _lib.check(_lib.load().lurk_hip_r1cs_cross_term_cached_dev(self._h, _lib.ptr(d_z2), *[_lib.ptr(x) for x in d_abc1], _lib.ptr(u1), *[_lib.ptr(x) for x in pv],
                                                                  _lib.ptr(rp), _lib.ptr(t), *[_lib.ptr(x) for x in abc2], _lib.ptr(s)))
        return t, abc2

    def close(self):
        if self._h:
            _lib.check(_lib.load().lurk_hip_r1cs_destroy(self._h))
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fold_vec(field_id: int, d_a, d_b, r_mont: np.ndarray, out=None, stream=None):
    """a + r b element-wise (RelaxedR1CSWitness::fold: W1 + r W2, E1 + r T); r = 4 x u64 Montgomery."""
    import torch

    n = d_a.shape[0]
    assert d_a.is_cuda and d_b.is_cuda and d_b.shape[0] == n
    if out is None:
        out = torch.empty_like(d_a)
    r = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.load().lurk_hip_fold_vec_dev(field_id, _lib.ptr(d_a), _lib.ptr(d_b), _lib.ptr(r), n, _lib.ptr(out), _lib.ptr(s)))
    return out


def fold_vecs(field_id: int, pairs, r_mont: np.ndarray, outs=None, stream=None):
    """Several folds a_k + r b_k under one r in ONE launch (``lurk_hip_fold_vecs_dev``, at most 8): pairs = [(d_a, d_b), ...];
    outs[k] may be d_a (in place).  Returns the outputs."""
    import torch

    n = len(pairs)
    if outs is None:
        outs = [torch.empty_like(a) for a, _ in pairs]
    r = np.ascontiguousarray(r_mont, dtype=np.uint64).reshape(4)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    pa = (ctypes.c_void_p * max(n, 1))(*[_lib.ptr(a) for a, _ in pairs])
    pb = (ctypes.c_void_p * max(n, 1))(*[_lib.ptr(b) for _, b in pairs])
    po = (ctypes.c_void_p * max(n, 1))(*[_lib.ptr(o) for o in outs])
    ns = (ctypes.c_size_t * max(n, 1))(*[a.shape[0] for a, _ in pairs])
    _lib.check(_lib.load().lurk_hip_fold_vecs_dev(field_id, n, pa, pb, ns, po, _lib.ptr(r), _lib.ptr(s)))
    return outs

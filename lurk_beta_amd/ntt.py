"""Radix-2 NTT over the Pasta fields on the GPU (no reference counterpart)."""
from __future__ import annotations

import numpy as np

from . import _lib


def ntt(field_id: int, data: np.ndarray, inverse: bool = False) -> np.ndarray:
    lib = _lib.load()
    a = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 4)
    n = a.shape[0]
    log_n = n.bit_length() - 1
    if n == 0 or (1 << log_n) != n:
        raise ValueError("length must be a power of two")
    _lib.check(lib.lurk_hip_ntt(field_id, _lib.ptr(a), log_n, int(inverse)))
    return a

"""Synthetic inputs generated directly in HBM.  Needs torch for device memory."""
from __future__ import annotations

from . import _lib


def scalars(field_id: int, stream_id: int, dist: int, n: int, first: int = 0, mont: bool = False, device="cuda"):
    import torch

    out = torch.empty((n, 4), dtype=torch.int64, device=device)
    lib = _lib.load()
    _lib.check(lib.lurk_hip_synth_scalars_dev(field_id, stream_id, dist, first, n, _lib.ptr(out), int(mont),
                                              _lib.ptr(torch.cuda.current_stream().cuda_stream)))
    return out


def bases(curve: int, n: int, first: int = 0, device="cuda"):
    import torch

    out = torch.empty((n, 8), dtype=torch.int64, device=device)
    lib = _lib.load()
    _lib.check(lib.lurk_hip_synth_bases_dev(curve, first, n, _lib.ptr(out), _lib.ptr(torch.cuda.current_stream().cuda_stream)))
    return out

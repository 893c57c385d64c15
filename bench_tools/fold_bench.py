#!/usr/bin/env python3
"""Times the fold kernels in isolation at step-circuit size: python bench_tools/fold_bench.py [rc]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import lurk_beta_amd as L
from lurk_beta_amd import _lib, synth

rc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_w, n_t, n_io = 9119 * rc, 11141 * rc, 2
F = L.FIELD_PALLAS_FQ
q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
lib = _lib.load()
shape = L.R1CSShape(F, n_t, n_w, n_io, *bench.synth_r1cs_shape(F, q, n_t, n_w, n_io))
info = shape.info()
d_z1 = synth.scalars(F, 1, 1, n_w + 1 + n_io, mont=True)
d_z2 = synth.scalars(F, 6, 1, n_w + 1 + n_io, mont=True)
d_t = torch.empty((n_t, 4), dtype=torch.int64, device="cuda")
r = np.array([1, 2, 3, 4], dtype=np.uint64)
for name, fn in (("cross_term", lambda: shape.cross_term(d_z1, d_z2, out=d_t)), ("multiply_vec", lambda: shape.multiply_vec(d_z1)),
                 ("fold_vec", lambda: L.fold_vec(F, d_z1, d_z2, r))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name:14s} {(time.perf_counter() - t0) / 20 * 1e3:8.3f} ms   nnz {sum(info['nnz'])}")

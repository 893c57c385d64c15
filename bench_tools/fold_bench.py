#!/usr/bin/env python3
"""Times the fold kernels in isolation at step-circuit size, HIP events around each launch, best and median of 30:
    python bench_tools/fold_bench.py [rc]          (LURK_HIP_LIB=<another build> for A/B runs of kernel variants)
r1cs_cross_term (six gathers per row), r1cs_cross_term_cached (round 6: z2's gathers alone + the cached products), the five-vector fold."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lurk_beta_amd as L
from bench_workloads.fold_step import synth_r1cs_shape
from lurk_beta_amd import synth

rc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_w, n_t, n_io = 8951 * rc + 64, 10973 * rc, 6
F = L.FIELD_PALLAS_FQ
q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
shape = L.R1CSShape(F, n_t, n_w, n_io, *synth_r1cs_shape(F, q, n_t, n_w, n_io))
nnz = sum(shape.info()["nnz"])
d_z1 = synth.scalars(F, 1, 1, n_w + 1 + n_io, mont=True)
d_z2 = synth.scalars(F, 6, 1, n_w + 1 + n_io, mont=True)
d_t = torch.empty((n_t, 4), dtype=torch.int64, device="cuda")
d_e = synth.scalars(F, 2, 0, n_t, mont=True)
abc1 = shape.multiply_vec(d_z1)
r = np.array([1, 2, 3, 4], dtype=np.uint64)
u1 = d_z1[n_w:n_w + 1].cpu().numpy().view(np.uint64)
t_c, abc2 = shape.cross_term_cached(d_z2, abc1, u1)
assert torch.equal(t_c, shape.cross_term(d_z1, d_z2)), "cached and six-gather cross terms differ"
outs = [torch.empty_like(d_z1), torch.empty_like(d_e)] + list(abc1)
cases = (("r1cs_cross_term", lambda: shape.cross_term(d_z1, d_z2, out=d_t), nnz * 72.0 + n_t * 44.0),
         ("r1cs_cross_term_cached", lambda: shape.cross_term_cached(d_z2, abc1, u1), nnz * 40.0 + n_t * 236.0),
         ("... + the cache's fold", lambda: shape.cross_term_cached(d_z2, abc1, u1, prev=abc2, r_prev_mont=r), nnz * 40.0 + n_t * (236.0 + 192.0)),
         ("fold_vecs (z, E)", lambda: L.fold_vecs(F, [(d_z1, d_z2), (d_e, d_t)], r, outs=outs[:2]), 96.0 * (n_w + 1 + n_io + n_t)),
         ("fold_vecs x5", lambda: L.fold_vecs(F, [(d_z1, d_z2), (d_e, d_t)] + list(zip(abc1, abc2)), r, outs=outs), 96.0 * (n_w + 1 + n_io + 4 * n_t)))
for name, fn, alg_bytes in cases:
    for _ in range(3):
        fn()
    ms = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    print(f"{name:24s} best {ms[0]:7.3f} ms  median {ms[15]:7.3f} ms   {alg_bytes / ms[15] / 1e6:7.1f} GB/s algorithmic = {alg_bytes / ms[15] / 8e7:5.1f} % of 8 TB/s   (rc {rc}, {n_t} rows, {nnz} non-zeros)")

"""Latency of one small commitment (synchronous, scalars resident in HBM) through the three forms of a resident key:
small (all multiples of all window bases, one launch), bucket (round-2 table key, 16-bit windows) and plain.
usage: python bench_tools/small_commit_probe.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lurk_beta_amd as L  # noqa: E402
from lurk_beta_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = []
for curve in (1, 0):
    sf = 1 - curve
    for n in (1 << 10, 1 << 13, 10_000, 1 << 14, 1 << 15, 1 << 16):
        d_b = synth.bases(curve, n)
        d_s = synth.scalars(sf, 1, 0, n, mont=True)
        row = {"curve": "vesta" if curve else "pallas", "n": n}
        res = {}
        for name, kw in (("small", dict(precompute=True)), ("bucket", dict(precompute=True, window_bits=16)), ("plain", dict())):
            t0 = time.perf_counter()
            key = L.CommitmentKey(curve, d_b, n=n, device=True, **kw)
            torch.cuda.synchronize()
            setup = time.perf_counter() - t0
            key.reserve(n, 2)
            for _ in range(5):
                r = key.commit_device(d_s, n, is_mont=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = key.commit_device(d_s, n, is_mont=True)
            dt = (time.perf_counter() - t0) / reps
            # two in flight
            t0 = time.perf_counter()
            for _ in range(reps // 2):
                key.submit_device(0, d_s, n, is_mont=True)
                key.submit_device(1, d_s, n, is_mont=True)
                key.wait(0)
                key.wait(1)
            dt2 = (time.perf_counter() - t0) / (reps // 2) / 2
            res[name] = L.point_to_affine(curve, r)
            row[name] = {"ms": round(dt * 1e3, 4), "ms_two_in_flight": round(dt2 * 1e3, 4), "setup_ms": round(setup * 1e3, 1)}
            key.close()
        row["agree"] = res["small"] == res["bucket"] == res["plain"]
        out.append(row)
        print(json.dumps(row), flush=True)

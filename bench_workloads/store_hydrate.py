"""--workload store_hydrate: StoreCore::hydrate_z_cache on the two DAG shapes that bound it."""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time

from .common import BENCH, ROOT, cpu_leg_omp, cpu_leg_threads


def store_hydrate_workload(args, lib, world, rank):
    """Store hydration (StoreCore::hydrate_z_cache, /root/reference/src/lem/store_core.rs:256-269) on the two DAG
    shapes that bound it: DEEP (a list of 400 distinct symbols: 25 wide levels of string / symbol hashing, then a spine of one cons per
    level) and WIDE (12 000 symbols under a balanced tree of conses: ~4 x 10^5 nodes, 22 levels).  A "step" is one whole hydration through
    lurk_hip_store_hydrate, host records in, host digests out (that IS the boundary: the store lives in host memory).  The CPU leg is the
    oracle's hydration (oracle.c: the same levels, every core) on the same DAGs; --verify compares every digest."""
    import numpy as np

    from lurk_beta_amd import store_hasher as SH
    from oracle import coracle as C

    F = 1
    out = None
    cpu_threads, cpu_info = cpu_leg_threads()
    cpu_leg_omp(cpu_threads)  # the CPU leg below runs with the thread count it reports
    for name, dag in (("deep", SH.list_dag(400)), ("wide", SH.wide_dag(12000))):
        rec, vals = SH.encode(dag)
        hashed = int((rec[:, 0] != 0).sum())
        for _ in range(max(1, args.warmup)):
            got, levels = SH.hydrate_records(F, rec, vals)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            got, levels = SH.hydrate_records(F, rec, vals)
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        r = {"ms_per_hydration": round(ms, 3), "nodes": len(dag), "hashed_nodes": hashed, "levels": levels,
             "value": round(hashed / ms / 1e3, 4), "unit": "M hashed nodes/s"}
        want = None
        if not args.no_cpu_baseline or args.verify:
            C.store_hydrate(F, rec[:64], vals)  # constants + thread pool
            t1 = time.perf_counter()
            want, _ = C.store_hydrate(F, rec, vals)
            dt = time.perf_counter() - t1
            r["cpu_baseline"] = {"value": round(hashed / dt / 1e6, 4), "unit": "M hashed nodes/s", "ms": round(dt * 1e3, 2), "cores": cpu_threads, **cpu_info, "kind": "port",
                                 "sample": "the same DAG, whole, oracle/oracle.c: orc_store_hydrate (level by level, OpenMP inside a level, plain-schedule Poseidon)"}
            r["speedup_vs_cpu_leg"] = round(dt * 1e3 / ms, 2)
        if args.verify:
            assert np.array_equal(got, want), f"{name}: digests differ from the oracle"
            r["verified"] = True
        if name == "deep":
            out = {"metric": "store hydration throughput (deep list DAG; the wide DAG is the `wide` sub-record)", "value": r["value"], "unit": r["unit"], "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_hydration"], "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "u32x8 (255-bit Montgomery, integer VALU) + 4x64 host limbs for narrow levels", "data": "synthetic",
                   "config": {"workload": f"lurk_hip_store_hydrate, {len(dag)}-node list DAG ({levels} levels), host records in / host digests out",
                              "verified": r.get("verified")},
                   "roofline": {"bound": "hbm", "kernel": "poseidon_wide_kernel", "achieved": round(hashed * 5 * 32 / (ms * 1e-3) / 1e9, 4), "peak": 8000.0,
                                "unit": "GB/s", "frac": round(hashed * 5 * 32 / (ms * 1e-3) / 8e12, 8), "traffic": None,
                                "note": "bound by the DAG's depth (one Poseidon dependency chain per level: ~0.14 ms on a GPU lane, ~25-50 us on a host core), "
                                        "not by HBM or VALU throughput: levels of at most 6 nodes are hashed by the library's host Poseidon"},
                   "deep": r}
        else:
            out["wide"] = r
    if "cpu_baseline" in out["deep"]:
        out["cpu_baseline"] = out["deep"]["cpu_baseline"]
    if rank == 0:
        print(json.dumps(out), flush=True)

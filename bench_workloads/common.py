"""Shared by the workloads: paths, the library profiler's per-kernel times, the cgroup CPU quota."""
from __future__ import annotations

import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def kernel_profile(lib, prefix):
    """(total ms, launches) of the profiler scopes whose name starts with `prefix` (HIP events on the launch stream, runtime.hip)"""
    from lurk_beta_amd import _lib

    tot, cnt = ctypes.c_double(), ctypes.c_uint64()
    _lib.check(lib.lurk_hip_profile_get(prefix.encode(), ctypes.byref(tot), ctypes.byref(cnt)))
    return tot.value, cnt.value


def cpu_quota_cores():
    """CPUs the cgroup lets this container use (cpu.max / cfs quota), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            q = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            per = float(fh.read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_leg_threads():
    """The thread count a cpu_baseline leg runs with, and the three numbers every leg reports beside its value (round 6: one helper
    for the headline and every sub-record).  A cgroup CPU quota caps the useful thread count whatever the host has (the GPU boxes
    show 256 logical CPUs under a 16-CPU quota): the OpenMP legs run with ceil(quota) threads, or every logical CPU without a quota.
    -> (threads, {"threads_used", "cpu_quota_cores", "host_cores"})"""
    import math

    quota = cpu_quota_cores()
    logical = os.cpu_count() or 1
    threads = max(1, min(logical, math.ceil(quota))) if quota else logical
    return threads, {"threads_used": threads, "cpu_quota_cores": quota, "host_cores": logical}


def cpu_leg_omp(threads):
    """Pin the oracle's OpenMP regions (oracle/oracle.c) to `threads` for a cpu_baseline leg."""
    from oracle import coracle as C

    C.lib().orc_set_num_threads(int(threads))

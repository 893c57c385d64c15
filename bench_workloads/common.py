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

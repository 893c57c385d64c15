"""Builds tests/host_harness/harness.cpp with g++ and exposes it via ctypes (CPU-only tests)."""
import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhost_harness.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        csrc = os.path.join(_HERE, "..", "..", "lurk_beta_amd", "csrc")
        srcs = [os.path.join(_HERE, "harness.cpp")] + glob.glob(os.path.join(csrc, "*.cuh")) + glob.glob(os.path.join(csrc, "*.hpp"))
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", _SO, os.path.join(_HERE, "harness.cpp")])
        _lib = ctypes.CDLL(_SO)
    return _lib

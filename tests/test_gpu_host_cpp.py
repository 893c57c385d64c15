"""Runs the C++ host-mirror parity program (tests/host_cpp/test_host.cpp: PoseidonCache, Trie and
CommitmentKey classes over the C ABI, reference KATs + group identities) on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror():
    exe = os.path.join(ROOT, "tests", "host_cpp", "test_host")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", exe + ".cpp", "-o", exe, "-L" + os.path.join(ROOT, "lurk_beta_amd"), "-llurk_hip",
                               "-Wl,-rpath,$ORIGIN/../../lurk_beta_amd"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror ok" in out.stdout

"""Every int-returning entry point of the C ABI called with all-null / all-zero arguments ON the GPU box: each must come back with
a non-zero code and a message (or succeed trivially), never crash the process - arecibo calls this library from rayon workers
and panics on the error code, a segfault would take the prover down without a trace.  One child process walks the whole ABI and
prints the name before each call, so a crash names its function."""
import ctypes
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import ctypes, sys
from lurk_beta_amd import _lib
lib = _lib.load()
for name, (res, args) in _lib.SIGNATURES.items():
    if res is not ctypes.c_int or name in ("lurk_hip_device_count", "lurk_hip_msm_multi_num_shards", "lurk_hip_abi_version"):  # these return a count / a revision
        continue
    vals = [None if (a in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(a, type) and issubclass(a, ctypes._Pointer))) else 0 for a in args]
    print("CALL", name, flush=True)
    rc = getattr(lib, name)(*vals)
    msg = (lib.lurk_hip_last_error() or b"").decode()
    print("RET", name, rc, msg[:60].replace("\n", " "), flush=True)
    assert rc == 0 or msg, name
print("DONE", flush=True)
"""


def test_null_arguments_never_crash(hip):
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    last = lines[-1] if lines else ""
    assert r.returncode == 0 and last == "DONE", f"child died (rc {r.returncode}) at: {last}\n{r.stderr[-400:]}"
    rets = {ln.split()[1]: int(ln.split()[2]) for ln in lines if ln.startswith("RET")}
    # destroy(NULL) is a no-op by contract; everything that needs an argument must have refused
    must_fail = [n for n in rets if not n.endswith("_destroy") and n not in (
        "lurk_hip_device_count", "lurk_hip_set_device", "lurk_hip_profile_enable", "lurk_hip_profile_reset", "lurk_hip_msm_oneshot_key_cache",
        "lurk_hip_msm_multi_num_shards", "lurk_hip_shake256", "lurk_hip_witness_blocks_dev",
        # all-zero arguments are an empty request (n = 0) for these: a no-op by contract
        "lurk_hip_ck_from_label_dev", "lurk_hip_ck_from_label_host", "lurk_hip_store_hydrate", "lurk_hip_synth_scalars_dev", "lurk_hip_synth_bases_dev",
        # set(NULL) restores the defaults, trim(NULL) just trims: both by contract (include/lurk_hip.h)
        "lurk_hip_ro_params_set", "lurk_hip_ck_params_set", "lurk_hip_scratch_trim")]
    silent = [n for n in must_fail if rets[n] == 0]
    assert not silent, f"accepted null arguments without an error: {silent}"

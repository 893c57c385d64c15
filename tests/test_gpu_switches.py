"""The library's environment switches: each one is read once per process, so each setting runs in its own
interpreter; whatever the setting, the commitments must be the oracle's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
import lurk_beta_amd as L
from oracle import coracle as C
n = 1 << 17
B = C.synth_bases(0, n)
key = L.CommitmentKey(0, B, precompute=True)          # bucket pipeline, 16-bit windows with the table
key.reserve(n, 3)
vecs = [C.synth_scalars(1, 40 + k, k %% 2, n) for k in range(3)]
dev = [torch.from_numpy(C.to_mont(1, v).view(np.int64)).cuda() for v in vecs]
torch.cuda.synchronize()
for rep in range(2):
    for k in range(3):
        key.submit_device(k, dev[k], n, is_mont=True)      # DEFAULT class: the persistent accumulation when the switch says so
    for k in range(3):
        assert L.point_to_affine(0, key.wait(k)) == C.jac_to_affine(0, C.msm_fast(0, B, vecs[k])), (rep, k)
# a step through the context (LURK_STEP_TRACE prints its phases)
A, Bm, Cm, z2 = C.synth_r1cs(1, 3000, 2500, 2, seed=9)
mont = lambda M: (M[0], M[1], C.to_mont(1, M[2]))
shape = L.R1CSShape(1, 3000, 2500, 2, mont(A), mont(Bm), mont(Cm))
k2 = L.CommitmentKey(0, B[:3000], precompute=True, window_bits=16)
ctx = L.FoldingContext(0, shape, k2)
cw, ct, r = ctx.step(C.to_mont(1, z2[:2500]), C.to_mont(1, z2[2501:]), 12345)
assert L.point_to_affine(0, cw) == C.jac_to_affine(0, C.msm_fast(0, B[:2500], z2[:2500]))
print("child ok")
'''


@pytest.mark.parametrize("env,expect_stderr", [
    ({}, None),
    ({"LURK_MSM_ACC_PERSISTENT": "0"}, None),                     # never the persistent accumulation
    ({"LURK_MSM_ACC_PERSISTENT": "2"}, None),                     # always (the default picks it from 24 x 2^20 sorted entries on)
    ({"LURK_MSM_PERSISTENT_MIN_MENTRIES": "1"}, None),            # the default rule with its threshold at 2^20 entries
    ({"LURK_MSM_ACC_PERSISTENT": "2", "LURK_MSM_PERSIST_WGS": "2", "LURK_MSM_MAX_ACC": "1"}, None),   # one two-wave accumulation at a time
    ({"LURK_MSM_TASK_TARGET": "1048576"}, None),                  # shorter accumulation tasks than the shape rule picks
    ({"LURK_MSM_BUCKET_DIRECT": "0"}, None),                      # short commitments (these: 2^21 entries) through the planned-task stages
    ({"LURK_MSM_ACC_PERSISTENT": "2", "LURK_MSM_MAX_ACC": "0"}, None),   # no limit on resident accumulations
    ({"LURK_MSM_ACC_PERSISTENT": "2", "LURK_MSM_PLACEMENT_LOG": "1"}, "accumulate placement"),
    ({"LURK_STEP_TRACE": "1"}, "[step]"),
])
def test_switch_settings_keep_the_results(hip, env, expect_stderr):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "child ok" in p.stdout
    if expect_stderr:
        assert expect_stderr in p.stderr, p.stderr[-500:]


KEY_CACHE_CHILD = r'''
import sys, time
import numpy as np
sys.path.insert(0, %r)
import lurk_beta_amd as L
from lurk_beta_amd import _lib
from oracle import coracle as C
n = 1 << 18
B = np.ascontiguousarray(C.synth_bases(0, n))
S = [C.synth_scalars(1, 70 + k, 0, n) for k in range(3)]
lib = _lib.load()
for k in range(3):   # pasta-msm's own symbol, host pointers: the second and third call find the key of the first in HBM
    out = np.zeros(12, dtype=np.uint64)
    lib.mult_pippenger_pallas(_lib.ptr(out), _lib.ptr(B), n, _lib.ptr(S[k]), False)
    assert L.point_to_affine(0, out) == C.jac_to_affine(0, C.msm_fast(0, B, S[k])), k
B2 = np.ascontiguousarray(C.synth_bases(0, n, first=n))
B[:] = B2            # same address, another key: the sampled positions differ, the cache must miss
out = np.zeros(12, dtype=np.uint64)
lib.mult_pippenger_pallas(_lib.ptr(out), _lib.ptr(B), n, _lib.ptr(S[0]), False)
assert L.point_to_affine(0, out) == C.jac_to_affine(0, C.msm_fast(0, B2, S[0]))
print("child ok")
'''


def test_oneshot_key_cache_from_the_environment(hip):
    """LURK_MSM_ONESHOT_KEY_CACHE=1 (include/lurk_hip.h): the opt-in key cache for callers that link pasta-msm's symbols unchanged."""
    e = dict(os.environ, LURK_MSM_ONESHOT_KEY_CACHE="1")
    p = subprocess.run([sys.executable, "-c", KEY_CACHE_CHILD % ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stderr[-2000:]

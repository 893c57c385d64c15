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
# the bucket reduction's launch mix (level pairs, single levels, wave butterflies: LURK_MSM_REDUCE_QUAD / _WAVE) depends on the window
# width and on the number of key spaces: a plain key (sixteen key spaces of 16-bit windows) and table keys of 18- and 20-bit windows
Bs, vs = B[:4096], C.synth_scalars(1, 90, 0, 4096)
want_s = C.jac_to_affine(0, C.msm_fast(0, Bs, vs))
for kw in (dict(precompute=False), dict(precompute=True, window_bits=18), dict(precompute=True, window_bits=20)):
    kk = L.CommitmentKey(0, Bs, **kw)
    assert L.point_to_affine(0, kk.commit(vs)) == want_s, kw
    kk.close()
# a step through the context (LURK_STEP_TRACE prints its phases)
A, Bm, Cm, z2 = C.synth_r1cs(1, 3000, 2500, 2, seed=9)
mont = lambda M: (M[0], M[1], C.to_mont(1, M[2]))
shape = L.R1CSShape(1, 3000, 2500, 2, mont(A), mont(Bm), mont(Cm))
k2 = L.CommitmentKey(0, B[:3000], precompute=True, window_bits=16)
ctx = L.FoldingContext(0, shape, k2)
cw, ct, r = ctx.step(C.to_mont(1, z2[:2500]), C.to_mont(1, z2[2501:]), 12345)
assert L.point_to_affine(0, cw) == C.jac_to_affine(0, C.msm_fast(0, B[:2500], z2[:2500]))
# two more steps with the NEXT instance staged ahead (its commitment in the class LURK_FOLD_STAGED_MODE names, FOLLOW by default) and the
# running products A z1, B z1, C z1 cached or not (LURK_FOLD_CACHED_PRODUCTS): commit(T) of a step depends on every fold before it
z1m, e1m = ctx.read()
z1, e1 = C.from_mont(1, z1m), C.from_mont(1, e1m)
mats = (A, Bm, Cm)
fresh = [np.concatenate([C.synth_scalars(1, 60 + k, 1, 2500), C.ints_to_limbs([1]), C.synth_scalars(1, 70 + k, 0, 2)]) for k in range(3)]
ctx.prefetch(C.to_mont(1, fresh[0][:2500]), 0)
for k in range(2):
    ctx.prefetch(C.to_mont(1, fresh[k + 1][:2500]), 0) if k == 0 else None
    cw, ct = ctx.begin_prefetched(C.to_mont(1, fresh[k][2501:]), [])
    u1 = C.limbs_to_ints(z1[2500:2501])[0]
    t = C.cross_term(1, *[C.spmv(1, *M, z1) for M in mats], *[C.spmv(1, *M, fresh[k]) for M in mats], u1, 1)
    assert L.point_to_affine(0, cw) == C.jac_to_affine(0, C.msm_fast(0, B[:2500], fresh[k][:2500])), k
    assert L.point_to_affine(0, ct) == C.jac_to_affine(0, C.msm_fast(0, B[:3000], t)), k
    rk = 0xABCDEF + k
    ctx.finish(C.to_mont(1, C.ints_to_limbs([rk])))
    z1, e1 = C.axpy(1, z1, fresh[k], rk), C.axpy(1, e1, t, rk)
gz, ge = ctx.read()
assert np.array_equal(C.from_mont(1, gz), z1) and np.array_equal(C.from_mont(1, ge), e1)
# the FOLLOW class on the bucket pipeline (2^22 sorted entries: the planned stages, not the one-launch form): a foreground commitment and
# one that follows it (sort at once, accumulation behind the first's), and a follower with nothing to follow
n2 = 1 << 18
B2 = C.synth_bases(0, n2)
kf = L.CommitmentKey(0, B2, precompute=True, window_bits=16)
kf.reserve(n2, 3)
v2 = [C.synth_scalars(1, 80 + k, 0, n2) for k in range(2)]
d2 = [torch.from_numpy(C.to_mont(1, v).view(np.int64)).cuda() for v in v2]
torch.cuda.synchronize()
want2 = [C.jac_to_affine(0, C.msm_fast(0, B2, v)) for v in v2]
kf.submit_device(2, d2[1], n2, is_mont=True, mode=3)
assert L.point_to_affine(0, kf.wait(2)) == want2[1]
for rep in range(2):
    kf.submit_device(1, d2[0], n2, is_mont=True, mode=1)
    kf.submit_device(0, d2[1], n2, is_mont=True, mode=3)
    assert L.point_to_affine(0, kf.wait(0)) == want2[1] and L.point_to_affine(0, kf.wait(1)) == want2[0], rep
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
    ({"LURK_FOLD_CACHED_PRODUCTS": "0"}, None),                   # the six-gather cross term of rounds 1-5
    ({"LURK_FOLD_STAGED_MODE": "2"}, None),                       # staged commitments in the BACKGROUND class (rounds 2-5)
    ({"LURK_FOLD_STAGED_MODE": "1"}, None),                       # ... in the foreground class
    ({"LURK_MSM_FOLLOW_WGS": "0"}, None),                         # a FOLLOW commitment's accumulation as the plain launch
    ({"LURK_MSM_FOLLOW_WGS": "1"}, None),
    ({"LURK_MSM_REDUCE_WAVE": "0"}, None),                        # one launch per reduction level
    ({"LURK_MSM_REDUCE_QUAD": "0"}, None),                        # no level pairs
    ({"LURK_MSM_REDUCE_QUAD": "0", "LURK_MSM_REDUCE_WAVE": "0"}, None),
    ({"LURK_MSM_FOLLOW_WGS": "3", "LURK_FOLD_CACHED_PRODUCTS": "0"}, None),
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

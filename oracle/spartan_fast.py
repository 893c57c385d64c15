"""CPU oracle: the SAME Spartan-style prover and verifier as oracle/spartan_ref.py - same statement, same five steps, same SHA3
transcript, hence the same proof element for element - with the vector work done by oracle/oracle.c (OpenMP) and the commitments by
oracle/msm_fast.c instead of Python integers, so that the oracle runs at 2^14 .. 2^20 rows (spartan_ref.py stops being usable around
2^7).  tests/test_oracle_spartan.py checks that the two produce identical proofs at small sizes.

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py's header).  Reference being restated: arecibo's RelaxedR1CSSNARK::{prove, verify} as
CompressedSNARK::prove runs it (/root/reference/src/proof/nova.rs:341-356); un-vendored, PARITY UNPINNED, not byte-compatible
(spartan_ref.py's header explains).

Conventions: vectors are (n, 4) uint64 canonical limb arrays; matrices are (indptr u64, indices u64, data (nnz, 4) canonical);
points are affine integer tuples or None at the interface (like spartan_ref), affine Montgomery limb arrays inside."""
from __future__ import annotations

import ctypes

import numpy as np

from . import coracle as C
from . import pyref as R
from .spartan_ref import Transcript, _sc_verify

_vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731


def _limbs(vals) -> np.ndarray:
    return C.ints_to_limbs([int(v) for v in vals])


def eq_evals(f: int, r: list[int]) -> np.ndarray:
    out = np.empty((1 << len(r), 4), dtype=np.uint64)
    rr = _limbs(r) if r else np.zeros((1, 4), dtype=np.uint64)
    C.lib().orc_eq_evals(f, _vp(rr), len(r), _vp(out))
    return out


def bind_top(f: int, t: np.ndarray, r: int) -> np.ndarray:
    out = np.empty((len(t) // 2, 4), dtype=np.uint64)
    C.lib().orc_bind_top(f, _vp(t), ctypes.c_size_t(len(t)), _vp(_limbs([r])), _vp(out))
    return out


def sumcheck_evals(f: int, tables: list[np.ndarray]) -> list[int]:
    degree = 3 if len(tables) == 4 else 2
    ptrs = (ctypes.c_void_p * len(tables))(*[t.ctypes.data for t in tables])
    out = np.zeros((3, 4), dtype=np.uint64)
    C.lib().orc_sumcheck_evals(f, degree, ptrs, ctypes.c_size_t(len(tables[0])), _vp(out))
    return C.limbs_to_ints(out[:degree])


def spmv_t(f: int, mat, v: np.ndarray, ncols: int) -> np.ndarray:
    indptr, indices, data = [np.ascontiguousarray(a, dtype=np.uint64) for a in mat]
    out = np.empty((ncols, 4), dtype=np.uint64)
    C.lib().orc_spmv_transposed(f, ctypes.c_size_t(len(indptr) - 1), _vp(indptr), _vp(indices), _vp(data), _vp(v), ctypes.c_size_t(ncols), _vp(out))
    return out


def sparse_mle(f: int, mat, eq_rx: np.ndarray, eq_ry: np.ndarray) -> int:
    indptr, indices, data = [np.ascontiguousarray(a, dtype=np.uint64) for a in mat]
    out = np.zeros((1, 4), dtype=np.uint64)
    C.lib().orc_sparse_mle(f, ctypes.c_size_t(len(indptr) - 1), _vp(indptr), _vp(indices), _vp(data), _vp(eq_rx), _vp(eq_ry), _vp(out))
    return C.limbs_to_ints(out)[0]


def fold_halves(f: int, v: np.ndarray, s_lo: int, s_hi: int) -> np.ndarray:
    out = np.empty((len(v) // 2, 4), dtype=np.uint64)
    C.lib().orc_fold_halves(f, _vp(v), ctypes.c_size_t(len(v)), _vp(_limbs([s_lo])), _vp(_limbs([s_hi])), _vp(out))
    return out


def points_fold_halves(curve: int, key: np.ndarray, s_lo: int, s_hi: int) -> np.ndarray:
    out = np.empty((len(key) // 2, 8), dtype=np.uint64)
    C.lib().orc_points_fold_halves(curve, _vp(key), ctypes.c_size_t(len(key)), _vp(_limbs([s_lo])), _vp(_limbs([s_hi])), _vp(out))
    return out


def lincomb(f: int, a: np.ndarray, b: np.ndarray, r: int) -> np.ndarray:
    return C.axpy(f, a, b, r)


def dot(f: int, a: np.ndarray, b: np.ndarray) -> int:
    return C.dot(f, np.ascontiguousarray(a), np.ascontiguousarray(b))


def _pad(v: np.ndarray, n: int) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    out[: len(v)] = v
    return out


def _aff(curve: int, jac: np.ndarray):
    a = C.jac_to_affine(curve, jac)
    return None if a == (0, 0) else a


def _commit(curve: int, key: np.ndarray, v: np.ndarray) -> np.ndarray:
    """Jacobian Montgomery sum v_i key_i (msm_fast.c above 256 points, the naive sum below)."""
    n = len(v)
    return C.msm_fast(curve, key[:n], v) if n > 256 else C.msm_naive(curve, key[:n], v)


def _point_mul_jac(curve: int, pt_aff_mont: np.ndarray, k: int) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    C.lib().orc_point_mul(curve, _vp(np.ascontiguousarray(pt_aff_mont)), _vp(_limbs([k])), _vp(out))
    return out


def _aff_mont(curve: int, pt) -> np.ndarray:
    """affine integer tuple / None -> (8,) affine Montgomery limbs"""
    if pt is None:
        return np.zeros(8, dtype=np.uint64)
    return C.to_mont(curve, _limbs([pt[0], pt[1]])).reshape(8)


def prove(curve_id: int, mats, num_cons: int, num_vars: int, X: list[int], ck: np.ndarray, comm_W, comm_E, u: int, W: np.ndarray, E: np.ndarray):
    """As spartan_ref.prove.  curve_id 0 / 1; ck: (>= N + 1, 8) affine Montgomery limbs, ck[N] is the argument's extra generator."""
    curve = "pallas" if curve_id == 0 else "vesta"
    f = 1 - curve_id
    q = R.CURVES[curve]["order"]
    ell_x, ell_y = num_cons.bit_length() - 1, num_vars.bit_length()
    N = max(num_cons, num_vars)
    ell = N.bit_length() - 1
    tr = Transcript(curve.encode())
    tr.absorb_point(b"comm_W", comm_W)
    tr.absorb_point(b"comm_E", comm_E)
    tr.absorb_scalars(b"uX", [u] + list(X))
    z = _pad(np.concatenate([W, _limbs([u] + list(X))]), 2 * num_vars)
    Az, Bz, Cz = [C.spmv(f, *M, z) for M in mats]
    tau = [tr.squeeze(b"t", q) for _ in range(ell_x)]
    uCzE = C.axpy(f, E, Cz, u)
    tables = [eq_evals(f, tau), Az, Bz, uCzE]
    polys_outer, r_x, claim = [], [], 0
    for _ in range(ell_x):
        e0, e2, e3 = sumcheck_evals(f, tables)
        poly = R.unipoly_from_evals(q, [e0, (claim - e0) % q, e2, e3])
        tr.absorb_scalars(b"p", poly)
        r = tr.squeeze(b"c", q)
        polys_outer.append(poly)
        r_x.append(r)
        claim = R.unipoly_eval(q, poly, r)
        tables = [bind_top(f, t, r) for t in tables]
    claim_Az, claim_Bz = C.limbs_to_ints(tables[1])[0], C.limbs_to_ints(tables[2])[0]
    eq_rx = eq_evals(f, r_x)
    claim_Cz, eval_E = dot(f, Cz, eq_rx), dot(f, E, eq_rx)
    tr.absorb_scalars(b"claims_outer", [claim_Az, claim_Bz, claim_Cz, eval_E])
    r = tr.squeeze(b"r", q)
    claim_inner = (claim_Az + r * claim_Bz + r * r * claim_Cz) % q
    eA, eB, eC = [spmv_t(f, M, eq_rx, 2 * num_vars) for M in mats]
    abc = C.axpy(f, C.axpy(f, eA, eB, r), eC, r * r % q)
    tables = [abc, z]
    polys_inner, r_y, claim = [], [], claim_inner
    for _ in range(ell_y):
        e0, e2 = sumcheck_evals(f, tables)
        poly = R.unipoly_from_evals(q, [e0, (claim - e0) % q, e2])
        tr.absorb_scalars(b"p", poly)
        rr = tr.squeeze(b"c", q)
        polys_inner.append(poly)
        r_y.append(rr)
        claim = R.unipoly_eval(q, poly, rr)
        tables = [bind_top(f, t, rr) for t in tables]
    eval_W = dot(f, W, eq_evals(f, r_y[1:]))
    tr.absorb_scalars(b"eval_W", [eval_W])
    P1, P2 = _pad(W, N), _pad(E, N)
    x1 = [0] * (ell - (ell_y - 1)) + r_y[1:]
    x2 = [0] * (ell - ell_x) + r_x
    rho = tr.squeeze(b"rho", q)
    pairs = [[eq_evals(f, x1), P1], [eq_evals(f, x2), P2]]
    coeffs = [1, rho]
    claim = (eval_W + rho * eval_E) % q
    polys_batch, r_z = [], []
    for _ in range(ell):
        e0 = e2 = 0
        for cf, (a, b) in zip(coeffs, pairs):
            s0, s2 = sumcheck_evals(f, [a, b])
            e0, e2 = (e0 + cf * s0) % q, (e2 + cf * s2) % q
        poly = R.unipoly_from_evals(q, [e0, (claim - e0) % q, e2])
        tr.absorb_scalars(b"p", poly)
        rr = tr.squeeze(b"c", q)
        polys_batch.append(poly)
        r_z.append(rr)
        claim = R.unipoly_eval(q, poly, rr)
        pairs = [[bind_top(f, a, rr), bind_top(f, b, rr)] for a, b in pairs]
    evals_batch = [C.limbs_to_ints(pairs[0][1])[0], C.limbs_to_ints(pairs[1][1])[0]]
    tr.absorb_scalars(b"evals_batch", evals_batch)
    gamma = tr.squeeze(b"gamma", q)
    a = C.axpy(f, P1, P2, gamma)
    r0 = tr.squeeze(b"ipa_r0", q)
    b = eq_evals(f, r_z)
    key = np.ascontiguousarray(ck[:N])
    ck_c2_jac = _point_mul_jac(curve_id, ck[N], r0)
    ck_c2 = C.jac_to_affine(curve_id, ck_c2_jac)
    ck_c2_mont = _aff_mont(curve_id, None if ck_c2 == (0, 0) else ck_c2)
    Ls, Rs = [], []
    while len(a) > 1:
        h = len(a) // 2
        c_L, c_R = dot(f, a[:h], b[h:]), dot(f, a[h:], b[:h])
        L = C.jac_add(curve_id, _commit(curve_id, key[h:], a[:h]), _point_mul_jac(curve_id, ck_c2_mont, c_L))
        Rr = C.jac_add(curve_id, _commit(curve_id, key[:h], a[h:]), _point_mul_jac(curve_id, ck_c2_mont, c_R))
        L, Rr = _aff(curve_id, L), _aff(curve_id, Rr)
        tr.absorb_point(b"L", L)
        tr.absorb_point(b"R", Rr)
        rr = tr.squeeze(b"r", q)
        ri = pow(rr, q - 2, q)
        Ls.append(L)
        Rs.append(Rr)
        a = fold_halves(f, a, rr, ri)
        b = fold_halves(f, b, ri, rr)
        key = points_fold_halves(curve_id, key, ri, rr)
    return dict(polys_outer=polys_outer, claims_outer=[claim_Az, claim_Bz, claim_Cz], eval_E=eval_E, polys_inner=polys_inner, eval_W=eval_W,
                polys_batch=polys_batch, evals_batch=evals_batch, ipa_L=Ls, ipa_R=Rs, ipa_a=C.limbs_to_ints(a)[0])


def verify(curve_id: int, mats, num_cons: int, num_vars: int, X: list[int], ck: np.ndarray, comm_W, comm_E, u: int, proof: dict) -> bool:
    """As spartan_ref.verify (the matrices are evaluated directly: the verifier of a preprocessing-free argument)."""
    curve = "pallas" if curve_id == 0 else "vesta"
    f = 1 - curve_id
    q = R.CURVES[curve]["order"]
    ell_x, ell_y = num_cons.bit_length() - 1, num_vars.bit_length()
    N = max(num_cons, num_vars)
    ell = N.bit_length() - 1
    tr = Transcript(curve.encode())
    tr.absorb_point(b"comm_W", comm_W)
    tr.absorb_point(b"comm_E", comm_E)
    tr.absorb_scalars(b"uX", [u] + list(X))
    tau = [tr.squeeze(b"t", q) for _ in range(ell_x)]

    def replay(polys):
        rs = []
        for poly in polys:
            tr.absorb_scalars(b"p", poly)
            rs.append(tr.squeeze(b"c", q))
        return rs

    if len(proof["polys_outer"]) != ell_x or len(proof["polys_inner"]) != ell_y or len(proof["polys_batch"]) != ell:
        return False
    r_x = replay(proof["polys_outer"])
    final = _sc_verify(q, 0, proof["polys_outer"], r_x)
    claim_Az, claim_Bz, claim_Cz = proof["claims_outer"]
    eval_E = proof["eval_E"]
    tau_rx = 1
    for t, rr in zip(tau, r_x):
        tau_rx = tau_rx * ((t * rr + (1 - t) * (1 - rr)) % q) % q
    if final is None or final != tau_rx * (claim_Az * claim_Bz - u * claim_Cz - eval_E) % q:
        return False
    tr.absorb_scalars(b"claims_outer", [claim_Az, claim_Bz, claim_Cz, eval_E])
    r = tr.squeeze(b"r", q)
    claim_inner = (claim_Az + r * claim_Bz + r * r * claim_Cz) % q
    r_y = replay(proof["polys_inner"])
    final = _sc_verify(q, claim_inner, proof["polys_inner"], r_y)
    eval_W = proof["eval_W"]
    eq_rx, eq_ry = eq_evals(f, r_x), eq_evals(f, r_y)
    abc = sum(pow(r, k, q) * sparse_mle(f, M, eq_rx, eq_ry) for k, M in enumerate(mats)) % q
    ux = _pad(_limbs([u] + list(X)), num_vars)
    eval_X = dot(f, ux, eq_evals(f, r_y[1:]))
    eval_z = ((1 - r_y[0]) * eval_W + r_y[0] * eval_X) % q
    if final is None or final != abc * eval_z % q:
        return False
    tr.absorb_scalars(b"eval_W", [eval_W])
    x1 = [0] * (ell - (ell_y - 1)) + r_y[1:]
    x2 = [0] * (ell - ell_x) + r_x
    rho = tr.squeeze(b"rho", q)
    r_z = replay(proof["polys_batch"])
    final = _sc_verify(q, (eval_W + rho * eval_E) % q, proof["polys_batch"], r_z)
    pw, pe = proof["evals_batch"]

    def eq_at(x, y):
        acc = 1
        for a, b in zip(x, y):
            acc = acc * ((a * b + (1 - a) * (1 - b)) % q) % q
        return acc

    if final is None or final != (eq_at(x1, r_z) * pw + rho * eq_at(x2, r_z) * pe) % q:
        return False
    tr.absorb_scalars(b"evals_batch", [pw, pe])
    gamma = tr.squeeze(b"gamma", q)
    comm_joint = R.ec_add(curve, comm_W, R.ec_mul(curve, gamma, comm_E))
    c = (pw + gamma * pe) % q
    r0 = tr.squeeze(b"ipa_r0", q)
    chal = []
    for L, Rr in zip(proof["ipa_L"], proof["ipa_R"]):
        tr.absorb_point(b"L", L)
        tr.absorb_point(b"R", Rr)
        chal.append(tr.squeeze(b"r", q))
    if len(chal) != ell:
        return False
    # InnerProductArgument::verify: P + sum_j (r_j^2 L_j + r_j^-2 R_j) == [a_hat] <s, ck> + [a_hat <s, b>] ck_c'
    ck_c2 = R.ec_mul(curve, r0, None if not ck[N].any() else C.affine_to_ints(curve_id, ck[N : N + 1])[0])
    P = R.ec_add(curve, comm_joint, R.ec_mul(curve, c, ck_c2))
    for rr, L, Rr in zip(chal, proof["ipa_L"], proof["ipa_R"]):
        ri = pow(rr, q - 2, q)
        P = R.ec_add(curve, P, R.ec_add(curve, R.ec_mul(curve, rr * rr % q, L), R.ec_mul(curve, ri * ri % q, Rr)))
    s = np.empty((N, 4), dtype=np.uint64)
    C.lib().orc_ipa_s_vector(f, _vp(_limbs(chal)), ell, _vp(s))
    ck_hat = _aff(curve_id, _commit(curve_id, np.ascontiguousarray(ck[:N]), s))
    b_hat = dot(f, s, eq_evals(f, r_z))
    a_hat = proof["ipa_a"]
    rhs = R.ec_add(curve, R.ec_mul(curve, a_hat, ck_hat), R.ec_mul(curve, a_hat * b_hat % q, ck_c2))
    return P == rhs


def synth_product_instance(f: int, nc: int, nv: int, nio: int, seed: int = 3):
    """A strictly satisfied instance of the kind bench.py's compress workload proves: rows 0 .. nc/2 - 1 are
    (sum of free variables, u, X) * (another sum) = own product variable, the other rows 0 * 0 = 0; columns over z = [W | u | X].
    Returns (A, B, C, W, X) with W (nv, 4) canonical and X a list of integers."""
    q = R.modulus(f)
    rng = np.random.default_rng(seed)
    rows_p = nc // 2
    nfree = nv - rows_p
    tab = _limbs([1, q - 1, 2, 3, 1 << 16, q - 7])

    def rand_mat():
        cnt = np.zeros(nc, dtype=np.uint64)
        cnt[:rows_p] = rng.integers(2, 5, rows_p)
        indptr = np.zeros(nc + 1, dtype=np.uint64)
        np.cumsum(cnt, out=indptr[1:])
        nnz = int(indptr[-1])
        cols = rng.integers(0, nfree + 1 + nio, nnz)
        cols = np.where(cols >= nfree, cols - nfree + nv, cols).astype(np.uint64)
        return indptr, cols, np.ascontiguousarray(tab[rng.integers(0, len(tab), nnz)])

    A, B = rand_mat(), rand_mat()
    cnt = np.zeros(nc, dtype=np.uint64)
    cnt[:rows_p] = 1
    ip = np.zeros(nc + 1, dtype=np.uint64)
    np.cumsum(cnt, out=ip[1:])
    Cm = (ip, (nfree + np.arange(rows_p)).astype(np.uint64), np.tile(tab[0], (rows_p, 1)))
    X = [R.uniform_fe(70 + seed, i, q) for i in range(nio)]
    z = np.zeros((2 * nv, 4), dtype=np.uint64)
    z[:nfree] = C.synth_scalars(f, 21, 1, nfree)
    z[nv : nv + 1 + nio] = _limbs([1] + X)
    az, bz = C.spmv(f, *A, z), C.spmv(f, *B, z)
    prod = np.empty((nc, 4), dtype=np.uint64)
    C.lib().orc_mul_canonical(f, _vp(az), _vp(bz), _vp(prod), ctypes.c_size_t(nc))
    z[nfree : nfree + rows_p] = prod[:rows_p]
    return A, B, Cm, np.ascontiguousarray(z[:nv]), X


# --------------------------------------------------------------------------------------------
# The batched variant: several relaxed R1CS instances of DIFFERENT shapes and sizes under one commitment key, one proof - the
# structure of arecibo's spartan::batched::BatchedRelaxedR1CSSNARK, which lurk-beta's SuperNova prover compresses with
# (/root/reference/src/proof/supernova.rs:110, 293-302): one outer (cubic) and one inner (quadratic) sum-check shared by all
# instances through random linear combinations, every instance's two evaluation claims (W_i, E_i) batched to ONE point, ONE
# inner-product-argument opening.  PARITY UNPINNED and not byte-compatible, exactly like the single-instance form above: instances
# of different sizes are zero-padded to the largest (padding at the high indices, so a padded evaluation is the small one times
# prod (1 - r_j) over the padding variables) where arecibo rescales claims, and the transcript is this file's.
# --------------------------------------------------------------------------------------------
def _pad_factor(q: int, r_pad: list[int]) -> int:
    acc = 1
    for r in r_pad:
        acc = acc * ((1 - r) % q) % q
    return acc


def prove_batched(curve_id: int, insts: list[dict], ck: np.ndarray):
    """insts[i]: dict(mats, num_cons, num_vars, X, u, W, E, comm_W, comm_E) with the conventions of prove().  ck: (>= N + 1, 8) affine
    Montgomery limbs, N = the largest num_cons / num_vars of the batch; ck[N] is the argument's extra generator."""
    curve = "pallas" if curve_id == 0 else "vesta"
    f = 1 - curve_id
    q = R.CURVES[curve]["order"]
    n = len(insts)
    ell_x = max(it["num_cons"] for it in insts).bit_length() - 1
    ell_y = max(it["num_vars"] for it in insts).bit_length()
    N = max(max(it["num_cons"], it["num_vars"]) for it in insts)
    ell = N.bit_length() - 1
    tr = Transcript(curve.encode() + b"/batched")
    tr.absorb_scalars(b"n", [n])
    for it in insts:
        tr.absorb_point(b"comm_W", it["comm_W"])
        tr.absorb_point(b"comm_E", it["comm_E"])
        tr.absorb_scalars(b"uX", [it["u"]] + list(it["X"]))
    tau = [tr.squeeze(b"t", q) for _ in range(ell_x)]
    rho_o = tr.squeeze(b"rho_outer", q)
    eq_tau = eq_evals(f, tau)
    zs, tabs = [], []
    for it in insts:
        nv = it["num_vars"]
        z = _pad(np.concatenate([it["W"], _limbs([it["u"]] + list(it["X"]))]), 2 * nv)
        Az, Bz, Cz = [C.spmv(f, *M, z) for M in it["mats"]]
        uCzE = C.axpy(f, it["E"], Cz, it["u"])
        zs.append(z)
        it["_Cz"] = Cz
        tabs.append([eq_tau, _pad(Az, 1 << ell_x), _pad(Bz, 1 << ell_x), _pad(uCzE, 1 << ell_x)])
    co = [pow(rho_o, i, q) for i in range(n)]
    polys_outer, r_x, claim = [], [], 0
    for _ in range(ell_x):
        e0 = e2 = e3 = 0
        for c, t in zip(co, tabs):
            a0, a2, a3 = sumcheck_evals(f, t)
            e0, e2, e3 = (e0 + c * a0) % q, (e2 + c * a2) % q, (e3 + c * a3) % q
        poly = R.unipoly_from_evals(q, [e0, (claim - e0) % q, e2, e3])
        tr.absorb_scalars(b"p", poly)
        r = tr.squeeze(b"c", q)
        polys_outer.append(poly)
        r_x.append(r)
        claim = R.unipoly_eval(q, poly, r)
        tabs = [[bind_top(f, t, r) for t in tt] for tt in tabs]
    eq_rx = eq_evals(f, r_x)
    claims_outer, evals_E = [], []
    for it, tt in zip(insts, tabs):
        nc = it["num_cons"]
        px = ell_x - (nc.bit_length() - 1)
        cA, cB = C.limbs_to_ints(tt[1])[0], C.limbs_to_ints(tt[2])[0]
        cC = dot(f, it["_Cz"], eq_rx[:nc])                         # padded evaluation: the tail of eq multiplies zeros
        eE = dot(f, it["E"], eq_evals(f, r_x[px:]))                # E_i itself at the sub-point (what the opening proves)
        claims_outer.append([cA, cB, cC])
        evals_E.append(eE)
    tr.absorb_scalars(b"claims_outer", [c for cl in claims_outer for c in cl] + evals_E)
    r = tr.squeeze(b"r", q)
    rho_i = tr.squeeze(b"rho_inner", q)
    pairs, claims_inner = [], []
    for it, z, cl in zip(insts, zs, claims_outer):
        nc, nv = it["num_cons"], it["num_vars"]
        eA, eB, eC = [spmv_t(f, M, np.ascontiguousarray(eq_rx[:nc]), 2 * nv) for M in it["mats"]]
        abc = C.axpy(f, C.axpy(f, eA, eB, r), eC, r * r % q)
        pairs.append([_pad(abc, 1 << ell_y), _pad(z, 1 << ell_y)])
        claims_inner.append((cl[0] + r * cl[1] + r * r * cl[2]) % q)
    ci = [pow(rho_i, i, q) for i in range(n)]
    claim = sum(c * e for c, e in zip(ci, claims_inner)) % q
    polys_inner, r_y = [], []
    for _ in range(ell_y):
        e0 = e2 = 0
        for c, (a, b) in zip(ci, pairs):
            s0, s2 = sumcheck_evals(f, [a, b])
            e0, e2 = (e0 + c * s0) % q, (e2 + c * s2) % q
        poly = R.unipoly_from_evals(q, [e0, (claim - e0) % q, e2])
        tr.absorb_scalars(b"p", poly)
        rr = tr.squeeze(b"c", q)
        polys_inner.append(poly)
        r_y.append(rr)
        claim = R.unipoly_eval(q, poly, rr)
        pairs = [[bind_top(f, a, rr), bind_top(f, b, rr)] for a, b in pairs]
    evals_W = []
    for it in insts:
        py = ell_y - it["num_vars"].bit_length()
        evals_W.append(dot(f, it["W"], eq_evals(f, r_y[py + 1:])))
    tr.absorb_scalars(b"evals_W", evals_W)
    # ---- 2 n evaluation claims -> one point -> one opening
    polys, points, claims = [], [], []
    for it, eW, eE in zip(insts, evals_W, evals_E):
        nc, nv = it["num_cons"], it["num_vars"]
        py, px = ell_y - nv.bit_length(), ell_x - (nc.bit_length() - 1)
        polys += [_pad(it["W"], N), _pad(it["E"], N)]
        points += [[0] * (ell - (nv.bit_length() - 1)) + r_y[py + 1:], [0] * (ell - (nc.bit_length() - 1)) + r_x[px:]]
        claims += [eW, eE]
    rho = tr.squeeze(b"rho", q)
    cb = [pow(rho, k, q) for k in range(2 * n)]
    bp = [[eq_evals(f, x), p] for x, p in zip(points, polys)]
    claim = sum(c * e for c, e in zip(cb, claims)) % q
    polys_batch, r_z = [], []
    for _ in range(ell):
        e0 = e2 = 0
        for c, (a, b) in zip(cb, bp):
            s0, s2 = sumcheck_evals(f, [a, b])
            e0, e2 = (e0 + c * s0) % q, (e2 + c * s2) % q
        poly = R.unipoly_from_evals(q, [e0, (claim - e0) % q, e2])
        tr.absorb_scalars(b"p", poly)
        rr = tr.squeeze(b"c", q)
        polys_batch.append(poly)
        r_z.append(rr)
        claim = R.unipoly_eval(q, poly, rr)
        bp = [[bind_top(f, a, rr), bind_top(f, b, rr)] for a, b in bp]
    evals_batch = [C.limbs_to_ints(b)[0] for _, b in bp]
    tr.absorb_scalars(b"evals_batch", evals_batch)
    gamma = tr.squeeze(b"gamma", q)
    a = polys[0]
    for k in range(1, 2 * n):
        a = C.axpy(f, a, polys[k], pow(gamma, k, q))
    r0 = tr.squeeze(b"ipa_r0", q)
    b = eq_evals(f, r_z)
    key = np.ascontiguousarray(ck[:N])
    ck_c2 = C.jac_to_affine(curve_id, _point_mul_jac(curve_id, ck[N], r0))
    ck_c2_mont = _aff_mont(curve_id, None if ck_c2 == (0, 0) else ck_c2)
    Ls, Rs = [], []
    while len(a) > 1:
        h = len(a) // 2
        c_L, c_R = dot(f, a[:h], b[h:]), dot(f, a[h:], b[:h])
        L = _aff(curve_id, C.jac_add(curve_id, _commit(curve_id, key[h:], a[:h]), _point_mul_jac(curve_id, ck_c2_mont, c_L)))
        Rr = _aff(curve_id, C.jac_add(curve_id, _commit(curve_id, key[:h], a[h:]), _point_mul_jac(curve_id, ck_c2_mont, c_R)))
        tr.absorb_point(b"L", L)
        tr.absorb_point(b"R", Rr)
        rr = tr.squeeze(b"r", q)
        ri = pow(rr, q - 2, q)
        Ls.append(L)
        Rs.append(Rr)
        a = fold_halves(f, a, rr, ri)
        b = fold_halves(f, b, ri, rr)
        key = points_fold_halves(curve_id, key, ri, rr)
    for it in insts:
        it.pop("_Cz", None)
    return dict(polys_outer=polys_outer, claims_outer=claims_outer, evals_E=evals_E, polys_inner=polys_inner, evals_W=evals_W, polys_batch=polys_batch,
                evals_batch=evals_batch, ipa_L=Ls, ipa_R=Rs, ipa_a=C.limbs_to_ints(a)[0])


def verify_batched(curve_id: int, insts: list[dict], ck: np.ndarray, proof: dict) -> bool:
    """insts[i]: dict(mats, num_cons, num_vars, X, u, comm_W, comm_E) (no witnesses)."""
    curve = "pallas" if curve_id == 0 else "vesta"
    f = 1 - curve_id
    q = R.CURVES[curve]["order"]
    n = len(insts)
    ell_x = max(it["num_cons"] for it in insts).bit_length() - 1
    ell_y = max(it["num_vars"] for it in insts).bit_length()
    N = max(max(it["num_cons"], it["num_vars"]) for it in insts)
    ell = N.bit_length() - 1
    if (len(proof["polys_outer"]), len(proof["polys_inner"]), len(proof["polys_batch"])) != (ell_x, ell_y, ell):
        return False
    if any(len(proof[k]) != n for k in ("claims_outer", "evals_E", "evals_W")) or len(proof["evals_batch"]) != 2 * n:
        return False
    tr = Transcript(curve.encode() + b"/batched")
    tr.absorb_scalars(b"n", [n])
    for it in insts:
        tr.absorb_point(b"comm_W", it["comm_W"])
        tr.absorb_point(b"comm_E", it["comm_E"])
        tr.absorb_scalars(b"uX", [it["u"]] + list(it["X"]))
    tau = [tr.squeeze(b"t", q) for _ in range(ell_x)]
    rho_o = tr.squeeze(b"rho_outer", q)

    def replay(polys):
        rs = []
        for poly in polys:
            tr.absorb_scalars(b"p", poly)
            rs.append(tr.squeeze(b"c", q))
        return rs

    def eq_at(x, y):
        acc = 1
        for a, b in zip(x, y):
            acc = acc * ((a * b + (1 - a) * (1 - b)) % q) % q
        return acc

    r_x = replay(proof["polys_outer"])
    final = _sc_verify(q, 0, proof["polys_outer"], r_x)
    tau_rx = eq_at(tau, r_x)
    want = 0
    for i, (it, (cA, cB, cC), eE) in enumerate(zip(insts, proof["claims_outer"], proof["evals_E"])):
        px = ell_x - (it["num_cons"].bit_length() - 1)
        eE_pad = _pad_factor(q, r_x[:px]) * eE % q
        want = (want + pow(rho_o, i, q) * tau_rx % q * (cA * cB - it["u"] * cC - eE_pad)) % q
    if final is None or final != want:
        return False
    tr.absorb_scalars(b"claims_outer", [c for cl in proof["claims_outer"] for c in cl] + list(proof["evals_E"]))
    r = tr.squeeze(b"r", q)
    rho_i = tr.squeeze(b"rho_inner", q)
    claim_inner = sum(pow(rho_i, i, q) * (cA + r * cB + r * r * cC) for i, (cA, cB, cC) in enumerate(proof["claims_outer"])) % q
    r_y = replay(proof["polys_inner"])
    final = _sc_verify(q, claim_inner, proof["polys_inner"], r_y)
    eq_rx, eq_ry = eq_evals(f, r_x), eq_evals(f, r_y)
    # the bound tables are abc_i'(r_y) z_i'(r_y), BOTH zero-padded: the truncated eq tables put the padding factors of r_x and r_y into
    # the sparse evaluation, z_i' = pad_y * z_i(sub-point) carries pad_y once more
    want = 0
    for i, (it, eW) in enumerate(zip(insts, proof["evals_W"])):
        nc, nv = it["num_cons"], it["num_vars"]
        py = ell_y - nv.bit_length()
        ex, ey = np.ascontiguousarray(eq_rx[:nc]), np.ascontiguousarray(eq_ry[: 2 * nv])
        abc = sum(pow(r, k, q) * sparse_mle(f, M, ex, ey) for k, M in enumerate(it["mats"])) % q
        rest = r_y[py + 1:]
        eval_X = dot(f, _pad(_limbs([it["u"]] + list(it["X"])), nv), eq_evals(f, rest))
        t = r_y[py]
        eval_z_pad = _pad_factor(q, r_y[:py]) * (((1 - t) * eW + t * eval_X) % q) % q
        want = (want + pow(rho_i, i, q) * abc % q * eval_z_pad) % q
    if final is None or final != want:
        return False
    tr.absorb_scalars(b"evals_W", list(proof["evals_W"]))
    points, claims, comms = [], [], []
    for it, eW, eE in zip(insts, proof["evals_W"], proof["evals_E"]):
        nc, nv = it["num_cons"], it["num_vars"]
        py, px = ell_y - nv.bit_length(), ell_x - (nc.bit_length() - 1)
        points += [[0] * (ell - (nv.bit_length() - 1)) + r_y[py + 1:], [0] * (ell - (nc.bit_length() - 1)) + r_x[px:]]
        claims += [eW, eE]
        comms += [it["comm_W"], it["comm_E"]]
    rho = tr.squeeze(b"rho", q)
    r_z = replay(proof["polys_batch"])
    final = _sc_verify(q, sum(pow(rho, k, q) * e for k, e in enumerate(claims)) % q, proof["polys_batch"], r_z)
    if final is None or final != sum(pow(rho, k, q) * eq_at(x, r_z) % q * e for k, (x, e) in enumerate(zip(points, proof["evals_batch"]))) % q:
        return False
    tr.absorb_scalars(b"evals_batch", list(proof["evals_batch"]))
    gamma = tr.squeeze(b"gamma", q)
    comm_joint, c = None, 0
    for k, (cm, e) in enumerate(zip(comms, proof["evals_batch"])):
        comm_joint = R.ec_add(curve, comm_joint, R.ec_mul(curve, pow(gamma, k, q), cm))
        c = (c + pow(gamma, k, q) * e) % q
    r0 = tr.squeeze(b"ipa_r0", q)
    chal = []
    for L, Rr in zip(proof["ipa_L"], proof["ipa_R"]):
        tr.absorb_point(b"L", L)
        tr.absorb_point(b"R", Rr)
        chal.append(tr.squeeze(b"r", q))
    if len(chal) != ell:
        return False
    ck_c2 = R.ec_mul(curve, r0, None if not ck[N].any() else C.affine_to_ints(curve_id, ck[N : N + 1])[0])
    P = R.ec_add(curve, comm_joint, R.ec_mul(curve, c, ck_c2))
    for rr, L, Rr in zip(chal, proof["ipa_L"], proof["ipa_R"]):
        ri = pow(rr, q - 2, q)
        P = R.ec_add(curve, P, R.ec_add(curve, R.ec_mul(curve, rr * rr % q, L), R.ec_mul(curve, ri * ri % q, Rr)))
    s = np.empty((N, 4), dtype=np.uint64)
    C.lib().orc_ipa_s_vector(f, _vp(_limbs(chal)), ell, _vp(s))
    ck_hat = _aff(curve_id, _commit(curve_id, np.ascontiguousarray(ck[:N]), s))
    b_hat = dot(f, s, eq_evals(f, r_z))
    a_hat = proof["ipa_a"]
    return P == R.ec_add(curve, R.ec_mul(curve, a_hat, ck_hat), R.ec_mul(curve, a_hat * b_hat % q, ck_c2))

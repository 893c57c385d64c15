"""CPU oracle: a complete Spartan-style SNARK for a relaxed R1CS instance - prover AND verifier - in plain Python integers
(SURVEY.md section 8 f3).

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py's header).

What it restates: the structure of arecibo's `spartan::snark::RelaxedR1CSSNARK::{prove, verify}` as `CompressedSNARK::prove` runs
it on each curve of the cycle (/root/reference/src/proof/nova.rs:341-356; arecibo is un-vendored, /root/reference/Cargo.toml:128):
  1. tau <- transcript; outer sum-check (cubic with additive term) of  sum_x eq(tau, x) (Az(x) Bz(x) - (u Cz(x) + E(x))) = 0
  2. claims Az(r_x), Bz(r_x), Cz(r_x), E(r_x); r <- transcript
  3. inner sum-check (quadratic) of  sum_y (A + r B + r^2 C)(r_x, y) z(y) = Az(r_x) + r Bz(r_x) + r^2 Cz(r_x)
  4. W(r_y[1:]); the two evaluation claims (W at r_y[1:], E at r_x) reduced to one point r_z by a batched quadratic sum-check
     of  sum_x sum_i rho^i eq(x_i, x) P_i(x)
  5. one inner-product-argument opening (oracle/pyref.py: ipa_prove) of P_1 + gamma P_2 at r_z.
PARITY UNPINNED and not byte-compatible: the transcript CONSTRUCTION is arecibo's Keccak256Transcript (oracle/keccak_transcript.py,
restated from its published source), but the labels and the order of what is absorbed are this restatement's own, vectors of
different lengths are zero-padded to a common power of two instead of arecibo's claim rescaling, and no proof bytes exist upstream.  What the oracle is for: (i) the device-assisted prover (lurk_beta_amd/spartan.py) must produce the SAME proof,
element for element, and (ii) `verify` below must accept it and reject tampered proofs - the size-independent property."""
from __future__ import annotations

from . import pyref as R
from .keccak_transcript import KeccakTranscript


class Transcript(KeccakTranscript):
    """arecibo's Keccak256Transcript (oracle/keccak_transcript.py) under this protocol's own label prefix: the construction is
    arecibo's, the labels and the order of the absorbed values are this restatement's (see the module docstring)."""

    def __init__(self, label: bytes):
        super().__init__(b"lurk-hip spartan v2" + label)


def mle_eval(p: int, table: list[int], point: list[int]) -> int:
    """Multilinear extension of `table` (len 2^len(point)) at `point`; point[0] <-> the most significant index bit."""
    eq = R.eq_evals(p, point)
    return sum(a * b for a, b in zip(table, eq)) % p


def _pad(v: list[int], n: int) -> list[int]:
    return list(v) + [0] * (n - len(v))


def _sc_verify(p: int, claim: int, polys: list[list[int]], challenges: list[int]) -> int | None:
    for poly, r in zip(polys, challenges):
        if (2 * poly[0] + sum(poly[1:])) % p != claim % p:  # p(0) + p(1)
            return None
        claim = R.unipoly_eval(p, poly, r)
    return claim


def sumcheck_prove_quad_batch(p: int, claims: list[int], pairs: list[tuple[list[int], list[int]]], coeffs: list[int], squeeze):
    """sum over pairs of coeff_i * sum_x A_i(x) B_i(x); one shared challenge per round from squeeze(round poly)."""
    pairs = [(list(a), list(b)) for a, b in pairs]
    claim = sum(c * e for c, e in zip(coeffs, claims)) % p
    polys, rs = [], []
    rounds = (len(pairs[0][0]) - 1).bit_length()
    for _ in range(rounds):
        e0 = e2 = 0
        for c, (a, b) in zip(coeffs, pairs):
            h = len(a) // 2
            s0 = sum(a[i] * b[i] for i in range(h)) % p
            s2 = sum((2 * a[h + i] - a[i]) * (2 * b[h + i] - b[i]) for i in range(h)) % p
            e0, e2 = (e0 + c * s0) % p, (e2 + c * s2) % p
        poly = R.unipoly_from_evals(p, [e0, (claim - e0) % p, e2])
        r = squeeze(poly)
        polys.append(poly)
        rs.append(r)
        claim = R.unipoly_eval(p, poly, r)
        pairs = [(R.bind_top(p, a, r), R.bind_top(p, b, r)) for a, b in pairs]
    return polys, rs, [(a[0], b[0]) for a, b in pairs], claim


def matrices_times(p: int, mats, z: list[int]):
    out = []
    for indptr, indices, data in mats:
        out.append([sum(data[k] * z[indices[k]] for k in range(indptr[i], indptr[i + 1])) % p for i in range(len(indptr) - 1)])
    return out


def matrices_transposed_times(p: int, mats, v: list[int], ncols: int):
    out = []
    for indptr, indices, data in mats:
        acc = [0] * ncols
        for i in range(len(indptr) - 1):
            for k in range(indptr[i], indptr[i + 1]):
                acc[indices[k]] = (acc[indices[k]] + data[k] * v[i]) % p
        out.append(acc)
    return out


def prove(curve: str, mats, num_cons: int, num_vars: int, X: list[int], ck: list, ck_c, comm_W, comm_E, u: int, W: list[int], E: list[int]):
    """mats = (A, B, C) as (indptr, indices, data) over columns z = [W | u | X | 0 ...] of length 2 num_vars; num_cons, num_vars powers of
    two, 1 + len(X) <= num_vars.  ck: >= max(num_cons, num_vars) affine points, ck_c one more.  Returns the proof (a dict)."""
    q = R.CURVES[curve]["order"]
    ell_x, ell_y = num_cons.bit_length() - 1, num_vars.bit_length()
    N = max(num_cons, num_vars)
    ell = N.bit_length() - 1
    tr = Transcript(curve.encode())
    tr.absorb_point(b"comm_W", comm_W)
    tr.absorb_point(b"comm_E", comm_E)
    tr.absorb_scalars(b"uX", [u] + list(X))
    z = _pad(list(W) + [u] + list(X), 2 * num_vars)
    Az, Bz, Cz = matrices_times(q, mats, z)
    tau = [tr.squeeze(b"t", q) for _ in range(ell_x)]
    uCzE = [(u * c + e) % q for c, e in zip(Cz, E)]
    chal_outer = []

    def sq_outer(j, poly):
        tr.absorb_scalars(b"p", poly)
        chal_outer.append(tr.squeeze(b"c", q))
        return chal_outer[-1]

    # round challenges depend on the round polynomials: run the oracle's prover round by round
    tables = [R.eq_evals(q, tau), Az, Bz, uCzE]
    polys_outer, claim = [], 0
    for j in range(ell_x):
        pl, _, _ = R.sumcheck_prove(q, claim, tables, [0])  # one round with a dummy challenge: take its polynomial ...
        poly = pl[0]
        r = sq_outer(j, poly)
        polys_outer.append(poly)
        claim = R.unipoly_eval(q, poly, r)
        tables = [R.bind_top(q, t, r) for t in tables]  # ... then bind with the real one
    r_x = chal_outer
    claim_Az, claim_Bz = tables[1][0], tables[2][0]
    claim_Cz, eval_E = mle_eval(q, Cz, r_x), mle_eval(q, E, r_x)
    tr.absorb_scalars(b"claims_outer", [claim_Az, claim_Bz, claim_Cz, eval_E])
    r = tr.squeeze(b"r", q)
    claim_inner = (claim_Az + r * claim_Bz + r * r * claim_Cz) % q
    eA, eB, eC = matrices_transposed_times(q, mats, R.eq_evals(q, r_x), 2 * num_vars)
    abc = [(a + r * b + r * r * c) % q for a, b, c in zip(eA, eB, eC)]
    tables = [abc, z]
    polys_inner, r_y, claim = [], [], claim_inner
    for j in range(ell_y):
        pl, _, _ = R.sumcheck_prove(q, claim, tables, [0])
        poly = pl[0]
        tr.absorb_scalars(b"p", poly)
        rr = tr.squeeze(b"c", q)
        polys_inner.append(poly)
        r_y.append(rr)
        claim = R.unipoly_eval(q, poly, rr)
        tables = [R.bind_top(q, t, rr) for t in tables]
    eval_W = mle_eval(q, W, r_y[1:])
    tr.absorb_scalars(b"eval_W", [eval_W])
    # ---- batch the two evaluation claims to one point
    P1, P2 = _pad(W, N), _pad(E, N)
    x1 = [0] * (ell - (ell_y - 1)) + r_y[1:]
    x2 = [0] * (ell - ell_x) + r_x
    rho = tr.squeeze(b"rho", q)

    def sq_batch(poly):
        tr.absorb_scalars(b"p", poly)
        return tr.squeeze(b"c", q)

    polys_batch, r_z, finals, _ = sumcheck_prove_quad_batch(q, [eval_W, eval_E], [(R.eq_evals(q, x1), P1), (R.eq_evals(q, x2), P2)], [1, rho], sq_batch)
    evals_batch = [finals[0][1], finals[1][1]]
    tr.absorb_scalars(b"evals_batch", evals_batch)
    gamma = tr.squeeze(b"gamma", q)
    joint = [(a + gamma * b) % q for a, b in zip(P1, P2)]
    r0 = tr.squeeze(b"ipa_r0", q)
    # the argument's challenges depend on L, R of the previous round: interleave
    a, b, key = list(joint), R.eq_evals(q, r_z), list(ck[:N])
    ck_c2 = R.ec_mul(curve, r0, ck_c)
    Ls, Rs = [], []
    while len(a) > 1:
        h = len(a) // 2
        c_L = sum(x * y for x, y in zip(a[:h], b[h:])) % q
        c_R = sum(x * y for x, y in zip(a[h:], b[:h])) % q
        L = R.ec_add(curve, R.msm_naive(curve, a[:h], key[h:]), R.ec_mul(curve, c_L, ck_c2))
        Rr = R.ec_add(curve, R.msm_naive(curve, a[h:], key[:h]), R.ec_mul(curve, c_R, ck_c2))
        tr.absorb_point(b"L", L)
        tr.absorb_point(b"R", Rr)
        rr = tr.squeeze(b"r", q)
        ri = pow(rr, q - 2, q)
        Ls.append(L)
        Rs.append(Rr)
        a = [(x * rr + ri * y) % q for x, y in zip(a[:h], a[h:])]
        b = [(x * ri + rr * y) % q for x, y in zip(b[:h], b[h:])]
        key = [R.ec_add(curve, R.ec_mul(curve, ri, l), R.ec_mul(curve, rr, k)) for l, k in zip(key[:h], key[h:])]
    return dict(polys_outer=polys_outer, claims_outer=[claim_Az, claim_Bz, claim_Cz], eval_E=eval_E, polys_inner=polys_inner, eval_W=eval_W,
                polys_batch=polys_batch, evals_batch=evals_batch, ipa_L=Ls, ipa_R=Rs, ipa_a=a[0])


def verify(curve: str, mats, num_cons: int, num_vars: int, X: list[int], ck: list, ck_c, comm_W, comm_E, u: int, proof: dict) -> bool:
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
    eq_rx, eq_ry = R.eq_evals(q, r_x), R.eq_evals(q, r_y)
    abc = 0
    for k, (indptr, indices, data) in enumerate(mats):
        acc = 0
        for i in range(len(indptr) - 1):
            for j in range(indptr[i], indptr[i + 1]):
                acc += data[j] * eq_rx[i] * eq_ry[indices[j]]
        abc = (abc + pow(r, k, q) * acc) % q
    eval_X = mle_eval(q, _pad([u] + list(X), num_vars), r_y[1:])
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
    return R.ipa_verify(curve, list(ck[:N]), ck_c, comm_joint, R.eq_evals(q, r_z), c, r0, chal, proof["ipa_L"], proof["ipa_R"], proof["ipa_a"])

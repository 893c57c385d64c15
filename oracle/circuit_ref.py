"""CPU oracle for the slot witnesses of a Lurk MultiFrame: the aux assignment a slot
contributes to the witness vector W, in the circuit's own allocation order.

TEST INFRASTRUCTURE ONLY (see oracle/pyref.py's header): nothing in the product imports this.

What the reference does: generate_slots_witnesses (/root/reference/src/lem/multiframe.rs:520-592) calls allocate_slot
(/root/reference/src/lem/circuit.rs:242-315) on a fresh WitnessCS per slot: the preimage elements are allocated first
(`AllocatedNum::alloc_infallible` per component, circuit.rs:258-287 / :297-305), then the image:
  * Hash4 / Hash6 / Hash8 / Commitment: `poseidon_hash` = neptune `circuit2::poseidon_hash_allocated` with the store's
    constants (circuit.rs:212-240);
  * BitDecomp: bellpepper `AllocatedNum::to_bits_le_strict` (circuit.rs:236-238).
Both gadgets live in un-vendored dependencies (neptune @ dev, bellpepper-core; /root/reference/Cargo.toml), so they are
restated here from their published algorithms:

  neptune circuit2 (optimized-static schedule, the one `Poseidon::hash()` runs):
    elements = [domain_tag (a constant Elt::Num)] + preimage (Elt::Allocated)
    first full round      l = e + pre_key ; aux l^2, l^4, l^5 + post_key      per element
    other full rounds     l = e           ; aux l^2, l^4, l^5 + post_key      (no post_key in the very last round)
    partial rounds        the same on element 0 only
    every S-box input after the first round is a linear combination (Elt::Num): nothing is allocated for it, the
    squaring / product constraints take the combination directly; the linear layers (dense MDS, the pre-sparse matrix
    after the last full round of the first half, one sparse matrix per partial round) allocate nothing;
    the digest elements[1] is allocated at the end (`ensure_allocated`): one more aux.
    => 3 * (t * R_F + R_P) + 1 aux after the preimage: hash3 265, hash4 289, hash6 337, hash8 388.
    keys = `compressed_round_constants` (preprocessing.rs: the first round's constants as they are; afterwards the constants of
    round r+1 pulled back through M^-1 to become post-S-box keys of round r; through the partial rounds only
    coordinate 0's key stays in its round, the rest is pushed one round earlier).
  bellpepper to_bits_le_strict: walk the bits of p - 1 from the top; a 1-bit allocates a boolean, a 0-bit closes the
    current run of 1-bits with a k-ary AND chain (one aux per AND, chained with the previous run's result) and
    allocates a boolean conditioned on it.

PINS (tests/test_oracle_circuit.py):
  * the four BitDecomp witness sizes the reference hardcodes - 298 (Pallas), 301 (Vesta), 354 (BN256), 364 (Grumpkin),
    /root/reference/src/lem/multiframe.rs:495-498 - are reproduced by the walk above (a strong check of the order);
  * the Poseidon digest aux equals the KAT-pinned hash (tests/golden/bn254_poseidon_kats.json);
  * every R1CS constraint the restated gadgets emit is satisfied by the assignment.
The Poseidon aux ORDER itself has no golden vector upstream: "parity unpinned beyond sizes, digests and satisfiability".
"""
from __future__ import annotations

from functools import lru_cache

from . import pyref as R

GRUMPKIN_FR = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # size pin only (multiframe.rs:498)


# ---- linear algebra mod p ---------------------------------------------------------------------------------------
def _mat_inv(m, p):
    n = len(m)
    a = [list(r) + [int(i == j) for j in range(n)] for i, r in enumerate(m)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % p)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], p - 2, p)
        a[c] = [x * inv % p for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % p for x, y in zip(a[r], a[c])]
    return [r[n:] for r in a]


def _mat_mul(a, b, p):
    return [[sum(a[i][k] * b[k][j] for k in range(len(b))) % p for j in range(len(b[0]))] for i in range(len(a))]


def _row_times(v, m, p):  # neptune's apply_matrix: row vector times matrix
    return [sum(v[i] * m[i][j] for i in range(len(v))) % p for j in range(len(m[0]))]


@lru_cache(maxsize=None)
def optimized_constants(field_id: int, arity: int):
    """neptune's PoseidonConstants for the optimized-static schedule: (compressed_round_constants, pre_sparse_matrix,
    [sparse matrices as (w_hat, v_rest)])."""
    p = R.modulus(field_id)
    t = arity + 1
    rf, rp = R.round_numbers(arity)
    h = rf // 2
    rc = R.round_constants(field_id, arity)
    m = [list(r) for r in R.mds_matrix(field_id, arity)]
    m_inv = _mat_inv(m, p)
    keys = lambda r: list(rc[r * t:(r + 1) * t])

    comp = list(keys(0))
    for i in range(h - 1):
        comp += _row_times(keys(i + 1), m_inv, p)
    partial_keys = []
    acc = keys(h + rp)
    for i in range(rp):
        inv = _row_times(acc, m_inv, p)
        partial_keys.append(inv[0])
        inv[0] = 0
        acc = [(a + b) % p for a, b in zip(keys(h + rp - i - 1), inv)]
    comp += _row_times(acc, m_inv, p)
    comp += list(reversed(partial_keys))
    for i in range(1, h):
        comp += _row_times(keys(i + h + rp), m_inv, p)
    assert len(comp) == rf * t + rp

    # sparse factorisation: M = M' M'', M' = diag(1, m_hat), M'' = [[m00, v], [w_hat, I]], w_hat = m_hat^-1 w
    def derive(mat):
        m_hat = [row[1:] for row in mat[1:]]
        w = [[mat[i][0]] for i in range(1, t)]
        w_hat = [r[0] for r in _mat_mul(_mat_inv(m_hat, p), w, p)]
        m_prime = [[1] + [0] * (t - 1)] + [[0] + list(r) for r in m_hat]
        m_dprime = [[mat[0][0]] + list(mat[0][1:])] + [[w_hat[i - 1]] + [int(i == j) for j in range(1, t)] for i in range(1, t)]
        return m_prime, m_dprime

    cur, sparse = m, []
    for _ in range(rp):
        m_prime, m_dprime = derive(cur)
        sparse.append(m_dprime)
        cur = _mat_mul(m, m_prime, p)
    sparse.reverse()
    pre_sparse = cur
    sp = [([row[0] for row in s], list(s[0][1:])) for s in sparse]  # (w_hat = column 0, v_rest = row 0 without its head)
    return tuple(comp), pre_sparse, sp


# ---- a tiny R1CS recorder ----------------------------------------------------------------------------------------
class WitnessCS:
    """aux[0..] in allocation order; variable 0 is ONE.  A linear combination is {var: coeff}; var = 1 + aux index."""

    def __init__(self, p):
        self.p = p
        self.aux: list[int] = []
        self.constraints: list[tuple[dict, dict, dict]] = []

    def alloc(self, v: int) -> int:
        self.aux.append(v % self.p)
        return len(self.aux)

    def enforce(self, a: dict, b: dict, c: dict):
        self.constraints.append((a, b, c))

    def value(self, lc: dict) -> int:
        z = [1] + self.aux
        return sum(z[k] * v for k, v in lc.items()) % self.p

    def unsatisfied(self) -> list[int]:
        return [i for i, (a, b, c) in enumerate(self.constraints) if self.value(a) * self.value(b) % self.p != self.value(c)]


def _lc_add(a: dict, b: dict, p: int, scale: int = 1) -> dict:
    out = dict(a)
    for k, v in b.items():
        out[k] = (out.get(k, 0) + v * scale) % p
    return out


def poseidon_hash_allocated(cs: WitnessCS, field_id: int, preimage_vars: list[int]) -> int:
    """neptune circuit2::poseidon_hash_allocated on already allocated preimage variables; returns the digest variable."""
    p = cs.p
    arity = len(preimage_vars)
    t = arity + 1
    rf, rp = R.round_numbers(arity)
    h = rf // 2
    comp, pre_sparse, sparse = optimized_constants(field_id, arity)
    mds = [list(r) for r in R.mds_matrix(field_id, arity)]
    elems: list[dict] = [{0: R.domain_tag(arity)}] + [{v: 1} for v in preimage_vars]
    off = 0

    def sbox(lc: dict, post):
        l = cs.value(lc)
        l2 = cs.alloc(l * l)
        cs.enforce(lc, lc, {l2: 1})
        l4 = cs.alloc(cs.aux[l2 - 1] ** 2)
        cs.enforce({l2: 1}, {l2: 1}, {l4: 1})
        l5v = (cs.aux[l4 - 1] * l + (post or 0)) % p
        l5 = cs.alloc(l5v)
        cs.enforce({l4: 1}, lc, _lc_add({l5: 1}, {0: (-(post or 0)) % p}, p))  # l4 * l = l5 - post
        return {l5: 1}

    def dense(mat):
        nonlocal elems
        out = []
        for j in range(t):
            acc: dict = {}
            for i in range(t):
                acc = _lc_add(acc, elems[i], p, mat[i][j])
            out.append(acc)
        elems = out

    def full_round(r, first, last):
        nonlocal off, elems
        pre = None
        if first:
            pre = comp[off:off + t]
            off += t
        post = None
        if first or not last:
            post = comp[off:off + t]
            off += t
        for i in range(t):
            lc = _lc_add(elems[i], {0: pre[i]}, p) if first else elems[i]
            elems[i] = sbox(lc, post[i] if post else None)
        dense(pre_sparse if r == h - 1 else mds)

    full_round(0, True, False)
    for r in range(1, h):
        full_round(r, False, False)
    for q in range(rp):
        elems[0] = sbox(elems[0], comp[off])
        off += 1
        w_hat, v_rest = sparse[q]
        new0: dict = {}
        for i in range(t):
            new0 = _lc_add(new0, elems[i], p, w_hat[i])
        elems = [new0] + [_lc_add(elems[j], elems[0], p, v_rest[j - 1]) for j in range(1, t)]
    for r in range(h - 1):
        full_round(h + rp + r, False, False)
    full_round(rf + rp - 1, False, True)
    assert off == len(comp)
    out = cs.alloc(cs.value(elems[1]))
    cs.enforce(elems[1], {0: 1}, {out: 1})
    return out


def bits_le_strict(cs: WitnessCS, var: int) -> list[int]:
    """bellpepper AllocatedNum::to_bits_le_strict; returns the bit variables, little-endian."""
    p = cs.p
    val = cs.aux[var - 1]
    pm1 = p - 1
    nbits_repr = 256
    result_be = []
    last_run = None
    current_run: list[int] = []
    found_one = False
    for i in reversed(range(nbits_repr)):
        b = (pm1 >> i) & 1
        a_bit = (val >> i) & 1
        found_one |= bool(b)
        if not found_one:
            assert a_bit == 0
            continue
        if b:
            v = cs.alloc(a_bit)
            cs.enforce({0: 1, v: p - 1}, {v: 1}, {})  # (1 - a) a = 0
            current_run.append(v)
            result_be.append(v)
        else:
            if current_run:
                if last_run is not None:
                    current_run.append(last_run)
                cur = current_run[0]
                for nxt in current_run[1:]:  # kary_and: AND them one by one
                    w = cs.alloc(cs.aux[cur - 1] & cs.aux[nxt - 1])
                    cs.enforce({cur: 1}, {nxt: 1}, {w: 1})
                    cur = w
                last_run = cur
                current_run = []
            v = cs.alloc(a_bit)
            cs.enforce({0: 1, last_run: p - 1, v: p - 1}, {v: 1}, {})  # (1 - must_be_false - a) a = 0
            result_be.append(v)
    assert not current_run
    lc: dict = {}
    coeff = 1
    for v in reversed(result_be):
        lc = _lc_add(lc, {v: coeff}, p)
        coeff = coeff * 2 % p
    cs.enforce(lc, {0: 1}, {var: 1})  # unpacking constraint
    return list(reversed(result_be))


SLOT_ARITY = {"hash4": 4, "hash6": 6, "hash8": 8, "commitment": 3}


def slot_witness(field_id: int, slot_type: str, preimage: list[int], modulus: int | None = None):
    """The aux assignment of one slot (allocate_slot on a fresh WitnessCS): (aux values, WitnessCS)."""
    p = modulus or R.modulus(field_id)
    cs = WitnessCS(p)
    pre = [cs.alloc(x) for x in preimage]
    if slot_type == "bit_decomp":
        assert len(pre) == 1
        bits_le_strict(cs, pre[0])
    else:
        assert len(pre) == SLOT_ARITY[slot_type]
        poseidon_hash_allocated(cs, field_id, pre)
    return list(cs.aux), cs


def slot_witness_size(field_id: int, slot_type: str, modulus: int | None = None) -> int:
    """compute_witness_size (/root/reference/src/lem/multiframe.rs:503-516)."""
    n = 1 if slot_type == "bit_decomp" else SLOT_ARITY[slot_type]
    return len(slot_witness(field_id, slot_type, [0] * n, modulus)[0])


# the step function's slot counts (/root/reference/src/lem/eval.rs:1960-1964) in the order generate_slots_witnesses walks them
STEP_SLOTS = (("hash4", 14), ("hash6", 0), ("hash8", 6), ("commitment", 1), ("bit_decomp", 3))


def frame_slot_block(field_id: int, preimages: dict[str, list[list[int]]]) -> list[int]:
    """The slot part of one frame's aux: slot witnesses back to back in (hash4, hash6, hash8, commitment, bit_decomp) order
    (synthesize_frame extends the frame's witness with them before anything else, circuit.rs:1429-1433)."""
    out: list[int] = []
    for typ, cnt in STEP_SLOTS:
        rows = preimages.get(typ, [])
        assert len(rows) == cnt, (typ, len(rows), cnt)
        for pre in rows:
            out += slot_witness(field_id, typ, pre)[0]
    return out

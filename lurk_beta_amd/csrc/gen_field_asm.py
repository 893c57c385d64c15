#!/usr/bin/env python3
"""Generates field_mul_asm.cuh: the 255-bit Montgomery product as ONE hand-scheduled gfx950 asm block
per field (and a squaring variant).

Why a generator: hipcc treats an `asm` statement as opaque - it pads one wait state after every
statement and cannot address the halves of a 64-bit operand - so a multiplier stitched from small
asm statements carries ~190 `s_nop`s and ~60 moves per product.  Here the whole column-wise
(finely-integrated product scanning) schedule is emitted explicitly:

  * accumulator  v[16:17] (64-bit mad destination) + v18 (carry count); m_0..m_7 in v19..v26.
    These are clobbered physical registers, so the text can name the pair and its halves.
  * every partial product is `v_mad_u64_u32 v[16:17], <carry pair>, x, y, v[16:17]`; its carry-out
    goes to one of four rotating SGPR pairs and is folded into v18 by a `v_addc_co_u32` emitted two
    instructions later - gfx950 requires two wait states between a VALU writing an SGPR and a VALU
    reading it as carry-in, and nothing pads hazards inside an asm block.  The first fold of a
    column is `v18 = 0 + 0 + carry`, so the carry word never needs zeroing.
  * per column: 2 moves to shift the accumulator down by one limb.
  * modulus limbs live in SGPRs (s_mov literals at the top; hoistable) - zero limbs are skipped,
    limb value 1 is the inline constant.

Run:  python gen_field_asm.py > field_mul_asm.cuh
"""
import sys

FIELDS = {
    "PallasFp": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
    "PallasFq": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    "Bn254Fr": 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
}

ACC_LO, ACC_HI, H = 16, 17, 18
M_BASE = 19            # m_i in v19..v26
P_SGPR_BASE = 4        # modulus limbs in s4..s11, -p^-1 in s12
INV_SGPR = 12
CARRY_PAIRS = [14, 16, 18, 20]


def limbs(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


class Emitter:
    def __init__(self):
        self.lines = []          # instruction texts
        self.writer_pos = {}     # sgpr pair -> index of the instruction that wrote it (VALU)
        self.free = list(CARRY_PAIRS)
        self.pending = []        # (pair, column_first_flag) carries not yet folded, in order

    def emit(self, text):
        self.lines.append(text)

    def wait_for(self, pair):
        """pad so that >= 2 instructions separate the VALU write of `pair` from the next instruction"""
        dist = len(self.lines) - 1 - self.writer_pos[pair]
        if dist < 2:
            self.emit(f"s_nop {2 - dist - 1}")

    def mad(self, x, y, first_fold_holder):
        # keep at most 2 unfolded carries behind the newest mad: fold the oldest before issuing
        while len(self.pending) > 2:
            self.fold(first_fold_holder)
        pair = self.free.pop(0)
        self.emit(f"v_mad_u64_u32 v[{ACC_LO}:{ACC_HI}], s[{pair}:{pair + 1}], {x}, {y}, v[{ACC_LO}:{ACC_HI}]")
        self.writer_pos[pair] = len(self.lines) - 1
        self.pending.append(pair)

    def fold(self, first_fold_holder):
        pair = self.pending.pop(0)
        self.wait_for(pair)
        if first_fold_holder[0]:
            self.emit(f"v_addc_co_u32_e64 v{H}, vcc, 0, 0, s[{pair}:{pair + 1}]")
            first_fold_holder[0] = False
        else:
            self.emit(f"v_addc_co_u32_e64 v{H}, vcc, 0, v{H}, s[{pair}:{pair + 1}]")
        self.free.append(pair)

    def drain(self, first_fold_holder):
        while self.pending:
            self.fold(first_fold_holder)


def gen_mul(name, modulus, square=False):
    p = limbs(modulus)
    inv = (-pow(modulus, -1, 1 << 32)) % (1 << 32)
    e = Emitter()
    # operand numbering: %0..%7 = t0..t7 (out), %8..%15 = a0..a7, %16..%23 = b0..b7 (square: b = a)
    T = lambda i: f"%{i}"
    A = lambda i: f"%{8 + i}"
    B = (lambda i: f"%{8 + i}") if square else (lambda i: f"%{16 + i}")
    M = lambda i: f"v{M_BASE + i}"

    def P(j):
        if p[j] == 1:
            return "1"
        return f"s{P_SGPR_BASE + j}"

    for j in range(8):
        if p[j] not in (0, 1):
            e.emit(f"s_mov_b32 s{P_SGPR_BASE + j}, 0x{p[j]:08x}")
    if inv != 0xFFFFFFFF:
        e.emit(f"s_mov_b32 s{INV_SGPR}, 0x{inv:08x}")

    for k in range(16):
        first_fold = [True]
        lo_i, hi_i = max(0, k - 7), min(k, 7)
        seq = []
        seq += [(A(i), B(k - i)) for i in range(lo_i, hi_i + 1)]
        seq += [(M(i), P(k - i)) for i in range(lo_i, min(k - 1, 7) + 1) if p[k - i] != 0]
        if k == 0:
            # accumulator starts empty: first product is a plain mad with a zero addend
            x, y = seq.pop(0)
            pair = e.free.pop(0)
            e.emit(f"v_mad_u64_u32 v[{ACC_LO}:{ACC_HI}], s[{pair}:{pair + 1}], {x}, {y}, 0")
            e.writer_pos[pair] = len(e.lines) - 1
            e.free.append(pair)          # cannot carry: 0 + product < 2^64
            e.emit(f"v_mov_b32 v{H}, 0")
            first_fold[0] = False
        for x, y in seq:
            e.mad(x, y, first_fold)
        if k < 8:
            if inv == 0xFFFFFFFF:
                e.emit(f"v_sub_u32 {M(k)}, 0, v{ACC_LO}")
            else:
                e.emit(f"v_mul_lo_u32 {M(k)}, v{ACC_LO}, s{INV_SGPR}")
            e.mad(M(k), P(0), first_fold)
        e.drain(first_fold)
        if first_fold[0]:
            # column without any fold (only k = 15): carry word of this column is zero
            e.emit(f"v_mov_b32 v{H}, 0")
        if k >= 8:
            e.emit(f"v_mov_b32 {T(k - 8)}, v{ACC_LO}")
        if k < 15:
            e.emit(f"v_mov_b32 v{ACC_LO}, v{ACC_HI}")
            e.emit(f"v_mov_b32 v{ACC_HI}, v{H}")
    return e.lines


def gen_redc(name, modulus):
    """Montgomery reduction of a 16-limb value v (as produced by the inner-product accumulator):
    t = (v + m*p) / 2^256, column-wise like the product, with v_k entering column k as v_k * 1."""
    p = limbs(modulus)
    inv = (-pow(modulus, -1, 1 << 32)) % (1 << 32)
    e = Emitter()
    T = lambda i: f"%{i}"
    V = lambda i: f"%{8 + i}"
    M = lambda i: f"v{M_BASE + i}"

    def P(j):
        return "1" if p[j] == 1 else f"s{P_SGPR_BASE + j}"

    for j in range(8):
        if p[j] not in (0, 1):
            e.emit(f"s_mov_b32 s{P_SGPR_BASE + j}, 0x{p[j]:08x}")
    if inv != 0xFFFFFFFF:
        e.emit(f"s_mov_b32 s{INV_SGPR}, 0x{inv:08x}")
    for k in range(16):
        first_fold = [True]
        seq = [(V(k), "1")]
        seq += [(M(i), P(k - i)) for i in range(max(0, k - 7), min(k - 1, 7) + 1) if p[k - i] != 0]
        if k == 0:
            x, y = seq.pop(0)
            pair = e.free.pop(0)
            e.emit(f"v_mad_u64_u32 v[{ACC_LO}:{ACC_HI}], s[{pair}:{pair + 1}], {x}, {y}, 0")
            e.writer_pos[pair] = len(e.lines) - 1
            e.free.append(pair)
            e.emit(f"v_mov_b32 v{H}, 0")
            first_fold[0] = False
        for x, y in seq:
            e.mad(x, y, first_fold)
        if k < 8:
            if inv == 0xFFFFFFFF:
                e.emit(f"v_sub_u32 {M(k)}, 0, v{ACC_LO}")
            else:
                e.emit(f"v_mul_lo_u32 {M(k)}, v{ACC_LO}, s{INV_SGPR}")
            e.mad(M(k), P(0), first_fold)
        e.drain(first_fold)
        if first_fold[0]:
            e.emit(f"v_mov_b32 v{H}, 0")
        if k >= 8:
            e.emit(f"v_mov_b32 {T(k - 8)}, v{ACC_LO}")
        if k < 15:
            e.emit(f"v_mov_b32 v{ACC_LO}, v{ACC_HI}")
            e.emit(f"v_mov_b32 v{ACC_HI}, v{H}")
    return e.lines


def cxx_redc(name, modulus):
    lines = gen_redc(name, modulus)
    nmad = sum(1 for l in lines if l.startswith("v_mad"))
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'"=&v"(t[{i}])' for i in range(8))
    ins = ", ".join(f'"v"(v[{i}])' for i in range(16))
    vclob = ", ".join(f'"v{r}"' for r in range(ACC_LO, M_BASE + 8))
    sclob = ", ".join(f'"s{r}"' for r in range(P_SGPR_BASE, CARRY_PAIRS[-1] + 2))
    return f"""// {name}: Montgomery reduction of a 16-limb value, {nmad} v_mad_u64_u32; result < 2^256, NOT yet in [0, p)
template <>
__device__ __forceinline__ void fe_redc16_asm<{name}>(uint32_t* t, const uint32_t* v) {{
    asm(
{body}
        : {outs}
        : {ins}
        : {vclob}, {sclob}, "vcc");
}}
"""


def cxx(name, modulus):
    lines = gen_mul(name, modulus)
    nmad = sum(1 for l in lines if l.startswith("v_mad"))
    nvalu = sum(1 for l in lines if l.startswith("v_"))
    nnop = sum(1 for l in lines if l.startswith("s_nop"))
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'"=&v"(t.l[{i}])' for i in range(8))
    ins = ", ".join([f'"v"(a.l[{i}])' for i in range(8)] + [f'"v"(b.l[{i}])' for i in range(8)])
    vclob = ", ".join(f'"v{r}"' for r in range(ACC_LO, M_BASE + 8))
    sclob = ", ".join(f'"s{r}"' for r in range(P_SGPR_BASE, CARRY_PAIRS[-1] + 2))
    return f"""// {name}: {nmad} v_mad_u64_u32, {nvalu} VALU instructions, {nnop} s_nop
template <>
__device__ __forceinline__ Fe<{name}> fe_mul_asm<{name}>(const Fe<{name}>& a, const Fe<{name}>& b) {{
    Fe<{name}> t;
    asm(
{body}
        : {outs}
        : {ins}
        : {vclob}, {sclob}, "vcc");
    fe_cond_sub<{name}>(t.l);
    return t;
}}
"""


def main():
    out = [
        "// field_mul_asm.cuh - GENERATED by gen_field_asm.py; do not edit.",
        "// One hand-scheduled gfx950 asm block per 255-bit Montgomery product (see the generator's docstring).",
        "#pragma once",
        "#if defined(__HIP_DEVICE_COMPILE__)",
        "namespace lurk {",
        "template <class P> __device__ __forceinline__ Fe<P> fe_mul_asm(const Fe<P>& a, const Fe<P>& b);",
        "template <class P> __device__ __forceinline__ void fe_redc16_asm(uint32_t* t, const uint32_t* v);",
        "",
    ]
    for name, mod in FIELDS.items():
        out.append(cxx(name, mod))
        out.append(cxx_redc(name, mod))
    out += ["}  // namespace lurk", "#else", "namespace lurk {",
            "// host pass: same name, portable arithmetic (kernels that name fe_mul_asm must still parse)",
            "template <class P> LURK_HD Fe<P> fe_mul_asm(const Fe<P>& a, const Fe<P>& b) { return fe_mul_fips<P>(a, b); }",
            "}  // namespace lurk", "#endif"]
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()

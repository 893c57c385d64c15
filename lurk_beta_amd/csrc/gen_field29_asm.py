#!/usr/bin/env python3
"""Generates field29_mul_asm.cuh: Montgomery product in radix 2^29 (9 limbs, R = 2^261) for gfx950.

Why a second representation: on gfx950 every carry fold (`v_addc_co_u32`, any form) issues at 4.1
cycles - nearly the cost of the `v_mad_u64_u32` (4.6) whose carry it folds (profiles/
r01_microbench_instr_rates.txt).  With 29-bit limbs a column of <= 9 + 6 partial products stays below
2^64, so the 64-bit mad accumulator never overflows: no carry folds at all, 135 mads instead of
104 mad+fold pairs for the Pasta moduli, and additions/subtractions become limb-wise (no carry chains,
values kept lazily in [0, 2^261)).

Emitted per field: one asm block  t = a*b / 2^261 mod p  (t limbs < 2^29, value < 2p+).
Contract: every limb of a and b < 2^30 (so a value may be the limb-wise sum of two normalised ones).
"""
import sys

FIELDS = {
    "PallasFp": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
    "PallasFq": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    "Bn254Fr": 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
}
W = 29
MASK = (1 << W) - 1
ACC_LO, ACC_HI = 16, 17
M_BASE = 18          # m_0..m_8 in v18..v26
P_SGPR_BASE = 4      # modulus limbs s4..s12
INV_SGPR = 13
MASK_VGPR = 27       # 2^29 - 1 in a VGPR: v_and with VGPR operands only issues at 2.3 cycles, with an SGPR at 4.1
DUMMY = 16           # s[16:17]: unused carry-out of the mads
P_VGPR_BASE = 28     # modulus limbs in v28..v36 when MOD_IN_VGPR (a mad with an SGPR multiplicand issues slower)
MOD_IN_VGPR = False
LAST_VGPR = P_VGPR_BASE + 8 if MOD_IN_VGPR else MASK_VGPR


def limbs29(x):
    return [(x >> (W * i)) & MASK for i in range(9)]


def gen_mul(modulus, square=False):
    """square=True: b is ignored; operands %18.. hold 2*a (limbs < 2^30) and the 36 cross products are issued once."""
    p = limbs29(modulus)
    inv = (-pow(modulus, -1, 1 << W)) % (1 << W)
    L = []
    T = lambda i: f"%{i}"
    A = lambda i: f"%{9 + i}"
    B = lambda i: f"%{18 + i}"
    M = lambda i: f"v{M_BASE + i}"

    def P(j):
        if p[j] == 1:
            return "1"
        return f"v{P_VGPR_BASE + j}" if MOD_IN_VGPR else f"s{P_SGPR_BASE + j}"

    for j in range(9):
        if p[j] not in (0, 1):
            if MOD_IN_VGPR:
                L.append(f"v_mov_b32 v{P_VGPR_BASE + j}, 0x{p[j]:08x}")
            else:
                L.append(f"s_mov_b32 s{P_SGPR_BASE + j}, 0x{p[j]:08x}")
    if inv != MASK:
        L.append(f"s_mov_b32 s{INV_SGPR}, 0x{inv:08x}")
    L.append(f"v_mov_b32 v{MASK_VGPR}, 0x{MASK:08x}")
    first = True
    for k in range(17):
        if square:
            # a_i * a_j (i < j) once, with the doubled copy B(j) = 2 a_j; the diagonal term with plain a_i
            prods = [(A(i), B(k - i)) for i in range(max(0, k - 8), min(k, 8) + 1) if i < k - i]
            if k % 2 == 0:
                prods.append((A(k // 2), A(k // 2)))
        else:
            prods = [(A(i), B(k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)]
        prods += [(M(i), P(k - i)) for i in range(max(0, k - 8), min(k - 1, 8) + 1) if p[k - i] != 0]
        for x, y in prods:
            src2 = "0" if first else f"v[{ACC_LO}:{ACC_HI}]"
            L.append(f"v_mad_u64_u32 v[{ACC_LO}:{ACC_HI}], s[{DUMMY}:{DUMMY + 1}], {x}, {y}, {src2}")
            first = False
        if k <= 8:
            # m_k = (-acc * p^-1) mod 2^29 ; acc += m_k * p_0  (clears the low 29 bits)
            if inv == MASK:
                L.append(f"v_sub_u32 {M(k)}, 0, v{ACC_LO}")
            else:
                L.append(f"v_mul_lo_u32 {M(k)}, v{ACC_LO}, s{INV_SGPR}")
            L.append(f"v_and_b32 {M(k)}, v{MASK_VGPR}, {M(k)}")
            L.append(f"v_mad_u64_u32 v[{ACC_LO}:{ACC_HI}], s[{DUMMY}:{DUMMY + 1}], {M(k)}, {P(0)}, v[{ACC_LO}:{ACC_HI}]")
        else:
            L.append(f"v_and_b32 {T(k - 9)}, v{MASK_VGPR}, v{ACC_LO}")
        if k < 16:
            L.append(f"v_lshrrev_b64 v[{ACC_LO}:{ACC_HI}], {W}, v[{ACC_LO}:{ACC_HI}]")
    # the top output limb takes everything that is left (value < 2^261 => fits)
    L.append(f"v_lshrrev_b64 v[{ACC_LO}:{ACC_HI}], {W}, v[{ACC_LO}:{ACC_HI}]")
    L.append(f"v_mov_b32 {T(8)}, v{ACC_LO}")
    return L, p, inv


def cxx_sqr(name, modulus):
    lines, p, inv = gen_mul(modulus, square=True)
    nmad = sum(1 for l in lines if l.startswith("v_mad"))
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'"=&v"(t.l[{i}])' for i in range(9))
    ins = ", ".join([f'"v"(a.l[{i}])' for i in range(9)] + [f'"v"(d.l[{i}])' for i in range(9)])
    vclob = ", ".join(f'"v{r}"' for r in range(ACC_LO, LAST_VGPR + 1))
    sclob = ", ".join(f'"s{r}"' for r in range(P_SGPR_BASE, DUMMY + 2))
    return f"""// {name} squaring: {nmad} v_mad_u64_u32 (cross products once, against the doubled operand)
template <>
__device__ __forceinline__ F29<{name}> f29_sqr_asm<{name}>(const F29<{name}>& a) {{
    F29<{name}> t, d;
#pragma unroll
    for (int i = 0; i < 9; i++) d.l[i] = a.l[i] << 1;
    asm(
{body}
        : {outs}
        : {ins}
        : {vclob}, {sclob}, "vcc");
    return t;
}}
"""


def cxx(name, modulus):
    lines, p, inv = gen_mul(modulus)
    nmad = sum(1 for l in lines if l.startswith("v_mad"))
    nvalu = sum(1 for l in lines if l.startswith("v_"))
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'"=&v"(t.l[{i}])' for i in range(9))
    ins = ", ".join([f'"v"(a.l[{i}])' for i in range(9)] + [f'"v"(b.l[{i}])' for i in range(9)])
    vclob = ", ".join(f'"v{r}"' for r in range(ACC_LO, LAST_VGPR + 1))
    sclob = ", ".join(f'"s{r}"' for r in range(P_SGPR_BASE, DUMMY + 2))
    return f"""// {name}: {nmad} v_mad_u64_u32, {nvalu} VALU instructions, no carry folds
template <>
__device__ __forceinline__ F29<{name}> f29_mul_asm<{name}>(const F29<{name}>& a, const F29<{name}>& b) {{
    F29<{name}> t;
    asm(
{body}
        : {outs}
        : {ins}
        : {vclob}, {sclob}, "vcc");
    return t;
}}
"""


def main():
    out = [
        "// field29_mul_asm.cuh - GENERATED by gen_field29_asm.py; do not edit.",
        "#pragma once",
        "#if defined(__HIP_DEVICE_COMPILE__)",
        "namespace lurk {",
        "template <class P> __device__ __forceinline__ F29<P> f29_mul_asm(const F29<P>& a, const F29<P>& b);",
        "template <class P> __device__ __forceinline__ F29<P> f29_sqr_asm(const F29<P>& a);",
        "",
    ]
    for name, mod in FIELDS.items():
        out.append(cxx(name, mod))
        out.append(cxx_sqr(name, mod))
    out += ["}  // namespace lurk", "#endif"]
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()

// field.cuh - 255-bit prime-field arithmetic for gfx950 (and for the host code of the library).
//
// Representation: 8 x 32-bit little-endian limbs in Montgomery form with R = 2^256.  This is
// byte-identical to the 4 x u64 `[u64;4]` Montgomery representation pasta_curves / halo2curves
// use, so commitment keys and witness vectors cross the C ABI without
// conversion (what arecibo hands to pasta-msm, /root/reference/src/proof/nova.rs:287-293).
//
// The same header compiles for the device (hipcc, gfx950) and for the host side of the library;
// tests/host_harness also builds it with g++ to check it against the oracle without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LURK_HD __host__ __device__ __forceinline__
#else
#define LURK_HD inline
#endif

namespace lurk {

// ------------------------------------------------------------------------------------------
// Field parameter packs.  All constants are constexpr so that the fully unrolled Montgomery
// reduction folds the zero / one limbs of the Pasta moduli (p = 2^254 + 126-bit tail:
// limbs 4..6 are 0, limb 0 is 1, -p^-1 mod 2^32 = -1) into fewer v_mad_u64_u32.
// ------------------------------------------------------------------------------------------
struct PallasFp {  // base field of Pallas = scalar field of Vesta
    static constexpr int ID = 0;
    static constexpr int NBITS = 255;
    static constexpr uint32_t INV = 0xffffffffu;
    LURK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0x00000001u, 0x992d30edu, 0x094cf91bu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
        return m[i];
    }
    LURK_HD static constexpr uint32_t r(int i) {
        constexpr uint32_t m[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return m[i];
    }
    LURK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};
        return m[i];
    }
    // 5^((p-1)/2^32): primitive 2^32-th root of unity, Montgomery form
    LURK_HD static constexpr uint32_t w32(int i) {
        constexpr uint32_t m[8] = {0xbad6dbf0u, 0xa28db849u, 0xd3b539dfu, 0x9083cd03u, 0x9dc8448eu, 0xfba6b9cau, 0x7b89c6dau, 0x3ec92874u};
        return m[i];
    }
};

struct PallasFq {  // scalar field of Pallas = base field of Vesta; LurkField for the Pallas cycle
    static constexpr int ID = 1;
    static constexpr int NBITS = 255;
    static constexpr uint32_t INV = 0xffffffffu;
    LURK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u};
        return m[i];
    }
    LURK_HD static constexpr uint32_t r(int i) {
        constexpr uint32_t m[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return m[i];
    }
    LURK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
        return m[i];
    }
    LURK_HD static constexpr uint32_t w32(int i) {
        constexpr uint32_t m[8] = {0x8c9942deu, 0x21807742u, 0x21b60494u, 0xcc495789u, 0xb2efbee2u, 0xac2e5d27u, 0x7f2db056u, 0x0b79fa89u};
        return m[i];
    }
};

struct Bn254Fr {  // BN254 scalar field: the field every golden vector of the reference is over
    static constexpr int ID = 2;
    static constexpr int NBITS = 254;
    static constexpr uint32_t INV = 0xefffffffu;
    LURK_HD static constexpr uint32_t mod(int i) {
        constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    LURK_HD static constexpr uint32_t r(int i) {
        constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    LURK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return m[i];
    }
    LURK_HD static constexpr uint32_t w32(int i) { return 0; }  // NTT is not offered over BN254
};

// ------------------------------------------------------------------------------------------
template <class P>
struct alignas(16) Fe {
    uint32_t l[8];
};

template <class P>
LURK_HD Fe<P> fe_zero() {
    Fe<P> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = 0;
    return z;
}
template <class P>
LURK_HD Fe<P> fe_one() {  // Montgomery 1
    Fe<P> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = P::r(i);
    return z;
}
template <class P>
LURK_HD Fe<P> fe_r2() {
    Fe<P> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.l[i] = P::r2(i);
    return z;
}
template <class P>
LURK_HD bool fe_is_zero(const Fe<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}
template <class P>
LURK_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// Carry chains.  On the device they are written with clang's add/sub-with-carry builtins: hipcc
// lowers them to v_add_co/v_addc_co chains, pads the gfx950 carry hazard (two wait states between a
// VALU carry write and its VALU read) and interleaves independent chains; the portable 64-bit form
// below compiles to v_lshl_add_u64 plus zero-extension moves and is ~4x slower on gfx950.
LURK_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned c;
    uint32_t r = __builtin_addc(a, b, carry, &c);
    carry = c;
    return r;
#else
    uint64_t x = (uint64_t)a + b + carry;
    carry = (uint32_t)(x >> 32);
    return (uint32_t)x;
#endif
}
LURK_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned c;
    uint32_t r = __builtin_subc(a, b, borrow, &c);
    borrow = c;
    return r;
#else
    uint64_t x = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(x >> 63);
    return (uint32_t)x;
#endif
}

// r = a - MOD if a >= MOD (a < 2*MOD)
template <class P>
LURK_HD void fe_cond_sub(uint32_t* t) {
    uint32_t d[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = subb32(t[i], P::mod(i), borrow);
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = borrow ? t[i] : d[i];
}

template <class P>
LURK_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = addc32(a.l[i], b.l[i], c);  // moduli are < 2^255: no carry out of limb 7
    fe_cond_sub<P>(r.l);
    return r;
}
template <class P>
LURK_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = subb32(a.l[i], b.l[i], borrow);
    uint32_t mask = 0u - borrow, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = addc32(r.l[i], P::mod(i) & mask, c);
    return r;
}
template <class P>
LURK_HD Fe<P> fe_neg(const Fe<P>& a) {
    return fe_sub<P>(fe_zero<P>(), a);
}
template <class P>
LURK_HD Fe<P> fe_dbl(const Fe<P>& a) {
    return fe_add<P>(a, a);
}

// ---- Montgomery product, portable form (host code of the library; device variant 0) --------
// CIOS, 32-bit limbs, 64-bit accumulate.  Valid "no extra carry word" form because every modulus
// here has its top limb < 2^31 - 1.
template <class P>
LURK_HD Fe<P> fe_mul_cios(const Fe<P>& a, const Fe<P>& b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t bi = b.l[i];
        uint64_t acc = (uint64_t)a.l[0] * bi + t[0];
        uint32_t m = (uint32_t)acc * P::INV;
        uint64_t red = (uint64_t)m * P::mod(0) + (uint32_t)acc;
        uint32_t c1 = (uint32_t)(acc >> 32), c2 = (uint32_t)(red >> 32);
#pragma unroll
        for (int j = 1; j < 8; j++) {
            acc = (uint64_t)a.l[j] * bi + t[j] + c1;
            c1 = (uint32_t)(acc >> 32);
            red = (uint64_t)m * P::mod(j) + (uint32_t)acc + c2;
            c2 = (uint32_t)(red >> 32);
            t[j - 1] = (uint32_t)red;
        }
        t[7] = c1 + c2;
    }
    fe_cond_sub<P>(t);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return r;
}

// ---- multiply-accumulate primitives for the column-wise (product scanning) device form ------
// acc64 += a*b with the carry-out captured as a lane mask in an SGPR pair; the carries of one
// column are folded into the third accumulator word afterwards (fold_carry).  gfx950 needs two
// wait states between a VALU writing an SGPR/VCC and a VALU reading it as carry-in, and hipcc does
// not pad hazards whose producer or consumer is inside an asm statement: deferring the v_addc to
// the end of the column provides the distance, column_fence() covers the short columns.
// Host: portable C++ with the same semantics (what tests/host_harness checks vs the oracle).
LURK_HD void mad_c(uint64_t& acc, uint64_t& carry, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "v"(b));
#else
    uint64_t p = (uint64_t)a * b;
    acc += p;
    carry = (acc < p) ? 1u : 0u;
#endif
}
// same with b a wave-uniform constant (modulus limb) held in an SGPR
LURK_HD void mad_ck(uint64_t& acc, uint64_t& carry, uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "s"(k));
#else
    mad_c(acc, carry, a, k);
#endif
}
LURK_HD void column_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1");
#endif
}
LURK_HD void fold_carry(uint32_t& hi, uint64_t carry) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(hi) : "s"(carry) : "vcc");
#else
    hi += (uint32_t)carry;
#endif
}

// Montgomery product a*b/R mod p.  Finely-integrated product scanning (column-wise) over 32-bit
// limbs: 64 v_mad_u64_u32 for a*b plus one per non-trivial modulus limb for the reduction (32 for
// the Pasta moduli, whose limbs 4..6 are 0 and limb 0 is 1; 64 for BN254).
template <class P>
LURK_HD Fe<P> fe_mul_fips(const Fe<P>& a, const Fe<P>& b) {
    uint32_t m[8], t[8];
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint64_t cm[16];
        int n = 0;
        const int i0 = k < 8 ? 0 : k - 7, i1 = k < 8 ? k : 7;
#pragma unroll
        for (int i = i0; i <= i1; i++) mad_c(lo, cm[n++], a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = i0; i <= (k < 8 ? k - 1 : 7); i++)
            if (P::mod(k - i) != 0) mad_ck(lo, cm[n++], m[i], P::mod(k - i));
        bool carry_in_one = false;
        uint32_t c0 = 0;
        if (k < 8) {
            c0 = (uint32_t)lo;
            m[k] = c0 * P::INV;
            if (P::mod(0) == 1 && P::INV == 0xffffffffu) {
                carry_in_one = true;  // m[k]*1 only clears the low word and carries iff it was non-zero
            } else {
                mad_ck(lo, cm[n++], m[k], P::mod(0));
            }
        }
        column_fence();
#pragma unroll
        for (int i = 0; i < n; i++) fold_carry(hi, cm[i]);
        if (k >= 8) t[k - 8] = (uint32_t)lo;  // after column 15 the overflow word is 0: result < 2p
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        if (carry_in_one) lo += (c0 != 0 ? 1u : 0u);
        hi = 0;
    }
    fe_cond_sub<P>(t);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return r;
}

// Host-side product (the library's host code: parameter generation, the <= 20-point MSM tail):
// 4 x 64-bit limbs, CIOS with unsigned __int128 - the same bytes as the 8 x 32-bit device form.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
template <class P>
inline Fe<P> fe_mul_host64(const Fe<P>& a, const Fe<P>& b) {
    typedef unsigned __int128 u128;
    uint64_t x[4], y[4], m[4], t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        m[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    }
    // -m^-1 mod 2^64 from the 32-bit constant by one Newton step
    uint64_t inv32 = P::INV;                      // -m^-1 mod 2^32
    uint64_t ninv = (uint64_t)0 - inv32;          // m^-1 mod 2^32 (as 64-bit, low half exact)
    ninv *= 2 - m[0] * ninv;                      // exact mod 2^64
    const uint64_t inv = (uint64_t)0 - ninv;
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)x[j] * y[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * inv;
        c = ((u128)q * m[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)q * m[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    // conditional subtraction
    uint64_t d[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 v = (u128)t[i] - m[i] - br;
        d[i] = (uint64_t)v;
        br = (v >> 64) & 1;
    }
    bool ge = t[4] != 0 || br == 0;
    Fe<P> r;
    for (int i = 0; i < 4; i++) {
        uint64_t v = ge ? d[i] : t[i];
        r.l[2 * i] = (uint32_t)v;
        r.l[2 * i + 1] = (uint32_t)(v >> 32);
    }
    return r;
}
#define LURK_HAVE_HOST64 1
#endif

// Device code calls the multiplier as a real function (arguments and result in VGPRs, no scratch):
// a 255-bit product is ~300 instructions, and kernels such as Poseidon (81 products per dense
// layer) or the XYZZ point addition would otherwise unroll into code far beyond the 64 KiB
// instruction cache.  LURK_MUL_IMPL=0 selects the portable CIOS form (compiler-scheduled).
// LURK_MUL_IMPL=2 (device default): the generated single-asm-block form (field_mul_asm.cuh);
// 1: the statement-per-mad form above; 0: portable CIOS.
#ifndef LURK_MUL_IMPL
#define LURK_MUL_IMPL 2
#endif
}  // namespace lurk
#include "field_mul_asm.cuh"
namespace lurk {
template <class P>
LURK_HD Fe<P> fe_mul_inline(const Fe<P>& a, const Fe<P>& b) {
#if LURK_MUL_IMPL == 0
    return fe_mul_cios<P>(a, b);
#elif LURK_MUL_IMPL == 2 && defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_asm<P>(a, b);
#elif !defined(__HIP_DEVICE_COMPILE__) && defined(LURK_HAVE_HOST64) && !defined(LURK_HOST_PORTABLE_MUL)
    return fe_mul_host64<P>(a, b);
#else
    return fe_mul_fips<P>(a, b);
#endif
}
#if defined(__HIPCC__)
template <class P>
__device__ __attribute__((noinline)) Fe<P> fe_mul_call(Fe<P> a, Fe<P> b) {
    return fe_mul_inline<P>(a, b);
}
#endif
template <class P>
LURK_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LURK_MUL_FORCE_INLINE)
    return fe_mul_call<P>(a, b);
#else
    return fe_mul_inline<P>(a, b);
#endif
}
// ---- inner product with ONE Montgomery reduction -----------------------------------------------
// sum_{i<T} a[i]*b[i] / R mod p.  Every partial product a_i[j]*b_i[l] is added to the private
// accumulator of column j+l (64-bit mad destination + carry word), so operands are consumed one at a
// time and the 16 columns are propagated and reduced once at the end: T*64 + 40 (Pasta) multiply-adds
// instead of T*104 - the dense and sparse layers of Poseidon are all inner products with constants.
// The unreduced result is < p*(T*p/R + 1) <= 3.25 p for T <= 9: two conditional subtractions.
template <class P>
LURK_HD void fe_cond_sub2(uint32_t* t) {  // t -= 2p if t >= 2p   (2p < 2^256 for every modulus here)
    uint32_t d[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t twop = (P::mod(i) << 1) | (i ? (P::mod(i - 1) >> 31) : 0u);
        d[i] = subb32(t[i], twop, borrow);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = borrow ? t[i] : d[i];
}

template <class P>
struct DotAcc {
    uint64_t col[15];
    uint32_t h[15];
};
template <class P>
LURK_HD void dot_init(DotAcc<P>& A) {
#pragma unroll
    for (int k = 0; k < 15; k++) {
        A.col[k] = 0;
        A.h[k] = 0;
    }
}
// A += a*b (64 multiply-adds, carries folded per column)
template <class P>
LURK_HD void dot_mac(DotAcc<P>& A, const Fe<P>& a, const Fe<P>& b) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t cm[8];
#pragma unroll
        for (int l = 0; l < 8; l++) mad_c(A.col[j + l], cm[l], a.l[j], b.l[l]);
        column_fence();
#pragma unroll
        for (int l = 0; l < 8; l++) fold_carry(A.h[j + l], cm[l]);
    }
}
// portable Montgomery reduction of a 16-limb value: t = (v + m*p) / 2^256 (< 2^256, not yet < p)
template <class P>
LURK_HD void fe_redc16_portable(uint32_t* t, const uint32_t* vin) {
    uint32_t v[17];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = vin[i];
    v[16] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t m = v[k] * P::INV;
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)m * P::mod(j) + v[k + j] + c;
            v[k + j] = (uint32_t)x;
            c = x >> 32;
        }
#pragma unroll
        for (int j = k + 8; j < 17; j++) {
            uint64_t x = (uint64_t)v[j] + c;
            v[j] = (uint32_t)x;
            c = x >> 32;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = v[8 + i];
}

// propagate the columns into a 16-limb integer (two interleaved carry chains), Montgomery-reduce it
// once, bring the result into [0, p).  The value is < T*p^2 < 2^512 for T <= 9, the quotient < 3.25 p.
template <class P, int T>
LURK_HD Fe<P> dot_finish(const DotAcc<P>& A) {
    // limb k receives: low word of column k, high word of column k-1, carry word of column k-2
    uint32_t x[16], v[16];
    uint32_t c1 = 0, c2 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t lo = k < 15 ? (uint32_t)A.col[k] : 0u;
        uint32_t hi = k >= 1 ? (uint32_t)(A.col[k - 1] >> 32) : 0u;
        x[k] = addc32(lo, hi, c1);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = addc32(x[k], k >= 2 ? A.h[k - 2] : 0u, c2);
    uint32_t t[8];
#if defined(__HIP_DEVICE_COMPILE__) && LURK_MUL_IMPL == 2
    fe_redc16_asm<P>(t, v);
#else
    fe_redc16_portable<P>(t, v);
#endif
    if (T > 3) fe_cond_sub2<P>(t);
    fe_cond_sub<P>(t);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return r;
}

template <class P>
LURK_HD Fe<P> fe_sqr(const Fe<P>& a) {
    return fe_mul<P>(a, a);
}

template <class P>
LURK_HD Fe<P> fe_to_mont(const Fe<P>& a) {
    return fe_mul<P>(a, fe_r2<P>());
}
template <class P>
LURK_HD Fe<P> fe_from_mont(const Fe<P>& a) {
    Fe<P> one = fe_zero<P>();
    one.l[0] = 1;
    return fe_mul<P>(a, one);
}
template <class P>
LURK_HD Fe<P> fe_from_u64(uint64_t v) {  // canonical small integer -> Montgomery
    Fe<P> x = fe_zero<P>();
    x.l[0] = (uint32_t)v;
    x.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont<P>(x);
}

// a^e, e given as 8 x 32-bit limbs (plain integer)
template <class P>
LURK_HD Fe<P> fe_pow(const Fe<P>& a, const uint32_t* e) {
    Fe<P> acc = fe_one<P>();
    for (int i = 255; i >= 0; i--) {
        acc = fe_sqr<P>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<P>(acc, a);
    }
    return acc;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// Host only: the inverse by the binary extended Euclidean algorithm (HAC 14.61 for an odd modulus) on 4 x 64-bit limbs - ~2 us where
// the exponentiation below takes ~24 (380 products of ~55 ns).  The host normalises every commitment it hands out and inverts every
// challenge of the opening argument: the step's and the provers' host tails are a few of these per round.  In: a R mod p (Montgomery
// form, canonical), out: a^-1 R mod p; 0 -> 0.  Not constant-time: its inputs are public (commitments, transcript challenges).
namespace host_inv {
struct U256 {
    uint64_t w[4];
};
inline bool is_one(const U256& a) { return a.w[0] == 1 && (a.w[1] | a.w[2] | a.w[3]) == 0; }
inline bool ge(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
inline uint64_t add(U256& a, const U256& b) {  // a += b, returns the carry
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)a.w[i] + b.w[i];
        a.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
inline void sub(U256& a, const U256& b) {  // a -= b (a >= b)
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = b.w[i], t = a.w[i] - bi, t2 = t - borrow;
        borrow = (uint64_t)(a.w[i] < bi) | (uint64_t)(t < borrow);
        a.w[i] = t2;
    }
}
inline void shr1(U256& a, uint64_t top) {  // a = (top : a) >> 1
    a.w[0] = (a.w[0] >> 1) | (a.w[1] << 63);
    a.w[1] = (a.w[1] >> 1) | (a.w[2] << 63);
    a.w[2] = (a.w[2] >> 1) | (a.w[3] << 63);
    a.w[3] = (a.w[3] >> 1) | (top << 63);
}
inline void halve_mod(U256& x, const U256& p) {  // x = x / 2 mod p (p odd, x < p)
    if (x.w[0] & 1) {
        const uint64_t c = add(x, p);
        shr1(x, c);
    } else {
        shr1(x, 0);
    }
}
inline void sub_mod(U256& x, const U256& y, const U256& p) {  // x = x - y mod p (x, y < p)
    if (ge(x, y)) {
        sub(x, y);
    } else {
        U256 t = p;
        sub(t, y);
        add(x, t);  // < p: no carry
    }
}
inline bool is_zero(const U256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
// a^-1 mod p for any 256-bit a (reduced first; a multiple of p gives 0), p an odd prime: total - it returns on every input, whatever a
// caller of the C ABI put into a point's coordinates
inline U256 inverse(U256 u, const U256& p) {
    while (ge(u, p)) sub(u, p);  // < 2^256 / p <= 5 rounds for the moduli here
    if (is_zero(u)) return u;
    U256 v = p, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
    while (!is_one(u) && !is_one(v)) {
        while (!(u.w[0] & 1)) {
            shr1(u, 0);
            halve_mod(x1, p);
        }
        while (!(v.w[0] & 1)) {
            shr1(v, 0);
            halve_mod(x2, p);
        }
        if (ge(u, v)) {
            sub(u, v);
            sub_mod(x1, x2, p);
        } else {
            sub(v, u);
            sub_mod(x2, x1, p);
        }
    }
    return is_one(u) ? x1 : x2;
}
}  // namespace host_inv
#endif

template <class P>
LURK_HD Fe<P> fe_inv(const Fe<P>& a) {  // a^(p-2); 0 -> 0
#if !defined(__HIP_DEVICE_COMPILE__)
    host_inv::U256 u, p;
    for (int i = 0; i < 4; i++) {
        u.w[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        p.w[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    }
    const host_inv::U256 x = host_inv::inverse(u, p);  // (a R)^-1 = a^-1 R^-1
    Fe<P> xi;
    for (int i = 0; i < 4; i++) {
        xi.l[2 * i] = (uint32_t)x.w[i];
        xi.l[2 * i + 1] = (uint32_t)(x.w[i] >> 32);
    }
    const Fe<P> r2 = fe_r2<P>();
    return fe_mul<P>(xi, fe_mul<P>(r2, r2));  // x * R^3 / R = a^-1 R
#else
    uint32_t e[8];
    uint32_t borrow = 2;
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)P::mod(i) - borrow;
        e[i] = (uint32_t)x;
        borrow = (uint32_t)(x >> 63);
    }
    return fe_pow<P>(a, e);
#endif
}
// the exponentiation everywhere (the device's form): the host harness holds the host's Euclidean inverse to it
template <class P>
LURK_HD Fe<P> fe_inv_pow(const Fe<P>& a) {
    uint32_t e[8];
    uint32_t borrow = 2;
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)P::mod(i) - borrow;
        e[i] = (uint32_t)x;
        borrow = (uint32_t)(x >> 63);
    }
    return fe_pow<P>(a, e);
}

// canonical integer comparison helper: is the canonical value >= MOD ?
template <class P>
LURK_HD bool fe_canonical_ge_mod(const uint32_t* v) {
    for (int i = 7; i >= 0; i--) {
        if (v[i] > P::mod(i)) return true;
        if (v[i] < P::mod(i)) return false;
    }
    return true;
}

}  // namespace lurk

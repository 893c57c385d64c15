// field29.cuh - the radix-2^29 arithmetic layer used INSIDE the hot kernels.
//
// gfx950 issues a carry fold (v_addc_co_u32, any form) at 4.1 cycles per wave-instruction - nearly the
// 4.6 of the v_mad_u64_u32 whose carry it folds (profiles/r01_microbench_instr_rates.txt).  With nine
// 29-bit limbs a whole column of partial products fits the 64-bit mad accumulator, so a product needs no
// carry folds: 135 mads instead of 104 mad+fold pairs (Pasta), 162 instead of 128 pairs (BN254).
//
// Representation: F29 = 9 limbs, Montgomery form with R' = 2^261 (value = x * 2^261 mod p, kept LAZILY
// in [0, 2^261), not reduced below p).  Limb bounds are part of every function's contract:
//   "tight"  every limb < 2^29   (products' outputs, f29_carry outputs, converted inputs)
//   "loose"  every limb < 2^31   (limb-wise sums / differences of tight values)
// f29_mul needs one operand tight and the other loose (or both with limbs < 2^30): 9 * 2^60 + 9 * 2^58
// + carry < 2^64.  Nothing at the C ABI changes: values cross into this layer with f29_from_mont256
// (x * 2^256 -> x * 2^261 is a 5-bit shift, folded into the limb repacking) and leave it with
// f29_to_mont256 (one product with 2^256, then a canonical reduction).
#pragma once
#include "field.cuh"

// LURK_F29_CHECK (host builds of the test harness only): every limb / accumulator bound the radix-2^29
// code relies on is asserted at run time.
#if defined(LURK_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#define F29_CHECKS_ACTIVE 1
#define F29_ASSERT_LIMBS(v, bits, what)                                                        \
    do {                                                                                       \
        for (int _i = 0; _i < 9; _i++)                                                         \
            if ((uint64_t)(v).l[_i] >> (bits)) { printf("F29 bound violated: %s limb %d = %u (>= 2^%d)\n", what, _i, (v).l[_i], bits); abort(); } \
    } while (0)
#define F29_ASSERT(cond, what)                                                                 \
    do {                                                                                       \
        if (!(cond)) { printf("F29 bound violated: %s\n", what); abort(); }                     \
    } while (0)
#define F29_ASSERT_TOP(v, bits, what)                                                          \
    do {                                                                                       \
        if ((uint64_t)(v).l[8] >> (bits)) { printf("F29 bound violated: %s top limb = %u (>= 2^%d)\n", what, (v).l[8], bits); abort(); } \
    } while (0)
#else
#define F29_ASSERT_LIMBS(v, bits, what) ((void)0)
#define F29_ASSERT_TOP(v, bits, what) ((void)0)
#define F29_ASSERT(cond, what) ((void)0)
#define F29_CHECKS_ACTIVE 0
#endif


namespace lurk {

template <class P>
struct F29 {
    uint32_t l[9];
};

constexpr uint32_t F29_MASK = (1u << 29) - 1u;

// modulus limbs in radix 2^29 (constexpr from the 8 x 32 description)
template <class P>
LURK_HD constexpr uint32_t f29_mod(int i) {
    // bits [29 i, 29 i + 29) of the 256-bit modulus
    const int bit = 29 * i, limb = bit >> 5, sh = bit & 31;
    uint64_t v = limb < 8 ? P::mod(limb) : 0u;
    if (limb + 1 < 8) v |= (uint64_t)P::mod(limb + 1) << 32;
    return (uint32_t)(v >> sh) & F29_MASK;
}
template <class P>
LURK_HD constexpr uint32_t f29_inv() {  // -p^-1 mod 2^29 (from the 32-bit constant)
    return P::INV & F29_MASK;
}

template <class P>
LURK_HD F29<P> f29_zero() {
    F29<P> z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.l[i] = 0;
    return z;
}

// one carry pass: loose -> tight limbs (value unchanged; the top limb keeps everything above bit 232)
template <class P>
LURK_HD F29<P> f29_carry(const F29<P>& a) {
    F29<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t x = a.l[i] + c;  // < 2^31 + 2^3
        r.l[i] = x & F29_MASK;
        c = x >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// limb-wise sum (tight + tight -> limbs < 2^30; tight + loose -> loose only if the caller knows the bound)
template <class P>
LURK_HD F29<P> f29_add(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// BIAS = 64p with its limbs re-balanced so that limbs 0..7 lie in [2^30, 2^30 + 2^29) and the top limb is
// ~2^28: limb_i += 2^30 and limb_{i+1} -= 2 (2^30 * 2^(29 i) = 2 * 2^(29 (i+1))).  Subtracting limb-wise any
// b with limbs < 2^30 and value < 2^260 then never underflows.
template <class P>
LURK_HD constexpr uint32_t f29_bias(int i) {
    uint64_t carry = 0;
    uint32_t limb = 0;
    for (int k = 0; k <= i; k++) {
        uint64_t x = (uint64_t)f29_mod<P>(k) * 64u + carry;
        limb = (uint32_t)(x & F29_MASK);
        carry = x >> 29;
        if (k == 8) limb = (uint32_t)x;  // top limb keeps the overflow (64p < 2^261)
    }
    uint32_t v = limb;
    if (i < 8) v += 1u << 30;  // lend to this limb ...
    if (i > 0) v -= 2u;        // ... what limb i-1 borrowed from it
    return v;
}
// a - b (mod p, lazily).  Contract: limbs(b) < 2^30, b < 2^260; a arbitrary with limbs < 2^29 (tight) or
// < 2^30: result limbs < 2^31 (loose), value < a + 2^260.
template <class P>
LURK_HD F29<P> f29_sub(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F29_ASSERT(f29_bias<P>(i) >= b.l[i], "f29_sub: subtrahend limb above the bias (value >= 64p or limbs not tight)");
        r.l[i] = a.l[i] + (f29_bias<P>(i) - b.l[i]);
    }
    return r;
}
template <class P>
LURK_HD F29<P> f29_dbl(const F29<P>& a) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1;
    return r;
}

// portable product (host / reference for the asm block): t = a*b / 2^261 mod p, limbs tight
template <class P>
LURK_HD F29<P> f29_mul_portable(const F29<P>& a, const F29<P>& b) {
    uint32_t m[9];
    F29<P> t;
    uint64_t acc = 0;
#if defined(LURK_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    {
        uint64_t ma = 0, mb = 0;
        for (int i = 0; i < 9; i++) { ma = a.l[i] > ma ? a.l[i] : ma; mb = b.l[i] > mb ? b.l[i] : mb; }
        // 9 products + 9 reduction terms (< 2^58 each) + the running carry must fit 64 bits
        F29_ASSERT((unsigned __int128)ma * mb * 9 + ((unsigned __int128)10 << 58) < ((unsigned __int128)1 << 64), "f29_mul column overflow");
    }
#endif
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k - 1 < 8 ? k - 1 : 8); i++) acc += (uint64_t)m[i] * f29_mod<P>(k - i);
        if (k <= 8) {
            m[k] = ((uint32_t)acc * f29_inv<P>()) & F29_MASK;
            acc += (uint64_t)m[k] * f29_mod<P>(0);
        } else {
            t.l[k - 9] = (uint32_t)acc & F29_MASK;
        }
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}

}  // namespace lurk
#include "field29_mul_asm.cuh"
namespace lurk {

template <class P>
LURK_HD F29<P> f29_mul(const F29<P>& a, const F29<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return f29_mul_asm<P>(a, b);
#else
    return f29_mul_portable<P>(a, b);
#endif
}
// a must be tight (the squaring block multiplies a by its doubled copy)
template <class P>
LURK_HD F29<P> f29_sqr(const F29<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return f29_sqr_asm<P>(a);
#else
    return f29_mul_portable<P>(a, a);
#endif
}

// a^(p-2): the inverse in the Montgomery(2^261) domain (0 -> 0).  a tight; result tight.  Square-and-multiply over the bits of
// p - 2, the same in every lane: 254 squarings and one product per set bit (the Pasta primes are 2^254 + a 126-bit tail, so ~65).
// On a lane this is ~0.15 ms where the 8 x 32 fe_inv takes ~0.6 (one wave per SIMD, dependent products).
template <class P>
LURK_HD F29<P> f29_invert(const F29<P>& a) {
    uint32_t e[8];
    uint32_t borrow = 2;
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)P::mod(i) - borrow;
        e[i] = (uint32_t)x;
        borrow = (uint32_t)(x >> 63);
    }
    int top = 255;
    while (top > 0 && !((e[top >> 5] >> (top & 31)) & 1u)) top--;
    F29<P> acc = a;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int b = top - 1; b >= 0; b--) {
        acc = f29_sqr<P>(acc);
        if ((e[b >> 5] >> (b & 31)) & 1u) acc = f29_mul<P>(acc, a);
    }
    return acc;
}

// ---- lazy inner products: sum_i a_i * b_i with ONE Montgomery reduction -------------------------------
// 17 unreduced 64-bit column sums take the 81 partial products of every term with no carry handling; 45
// products of tight limbs (< 2^58 each) fit a column together with the reduction's own terms.
template <class P>
struct Dot29 {
    uint64_t c[17];
#if defined(LURK_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    unsigned __int128 shadow[17];  // exact column sums: the 64-bit ones must never wrap
#endif
};
template <class P>
LURK_HD void dot29_init(Dot29<P>& A) {
#pragma unroll
    for (int k = 0; k < 17; k++) A.c[k] = 0;
#if defined(LURK_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < 17; k++) A.shadow[k] = 0;
#endif
}
// A += a * b  (both tight: 81 products of < 2^58)
template <class P>
LURK_HD void dot29_mac(Dot29<P>& A, const F29<P>& a, const F29<P>& b) {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) A.c[i + j] += (uint64_t)a.l[i] * b.l[j];
#if defined(LURK_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) A.shadow[i + j] += (unsigned __int128)a.l[i] * b.l[j];
    // room for the reduction's own terms (9 products < 2^58 + carry) must remain
    for (int k = 0; k < 17; k++) F29_ASSERT(A.shadow[k] + ((unsigned __int128)10 << 58) < ((unsigned __int128)1 << 64), "dot29 column overflow");
#endif
}
// push every column's excess above 29 bits into the next column (value unchanged)
template <class P>
LURK_HD void dot29_carry(Dot29<P>& A) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A.c[k + 1] += A.c[k] >> 29;
        A.c[k] &= F29_MASK;
    }
#if defined(LURK_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < 17; k++) A.shadow[k] = A.c[k];
#endif
}
// t = A / 2^261 mod p, tight, value < A / 2^261 + p
template <class P>
LURK_HD F29<P> dot29_finish(const Dot29<P>& A) {
    uint32_t m[9];
    F29<P> t;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        acc += A.c[k];
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k - 1 < 8 ? k - 1 : 8); i++) acc += (uint64_t)m[i] * f29_mod<P>(k - i);
        if (k <= 8) {
            m[k] = ((uint32_t)acc * f29_inv<P>()) & F29_MASK;
            acc += (uint64_t)m[k] * f29_mod<P>(0);
        } else {
            t.l[k - 9] = (uint32_t)acc & F29_MASK;
        }
        acc >>= 29;
    }
    t.l[8] = (uint32_t)acc;
    return t;
}

// 8 x 32 Montgomery(2^256) -> 9 x 29 Montgomery(2^261): value * 32, i.e. limbs of (x << 5); tight.
template <class P>
LURK_HD F29<P> f29_from_mont256(const Fe<P>& x) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i - 5;  // bit of x where limb i starts (may be negative for i = 0)
        uint64_t v;
        if (bit < 0) {
            v = (uint64_t)x.l[0] << 5;
        } else {
            const int limb = bit >> 5, sh = bit & 31;
            v = limb < 8 ? x.l[limb] : 0u;
            if (limb + 1 < 8) v |= (uint64_t)x.l[limb + 1] << 32;
            v >>= sh;
        }
        r.l[i] = (uint32_t)v & F29_MASK;
    }
    return r;
}
// plain 8 x 32 integer -> 9 x 29 limbs of the same integer (no shift); tight
template <class P>
LURK_HD F29<P> f29_from_plain(const uint32_t* x) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, limb = bit >> 5, sh = bit & 31;
        uint64_t v = limb < 8 ? x[limb] : 0u;
        if (limb + 1 < 8) v |= (uint64_t)x[limb + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & F29_MASK;
    }
    return r;
}
// tight 9 x 29 limbs (value < 2^256 after reduction) -> 8 x 32 words
template <class P>
LURK_HD void f29_pack(const F29<P>& a, uint32_t* out) {
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i = bit / 29, sh = bit % 29;
        uint64_t v = (uint64_t)a.l[i] >> sh;
        if (i + 1 < 9) v |= (uint64_t)a.l[i + 1] << (29 - sh);
        if (i + 2 < 9) v |= (uint64_t)a.l[i + 2] << (58 - sh);
        out[w] = (uint32_t)v;
    }
}
// 2^256 mod p as plain limbs: multiplying by it in the 2^261 domain divides by 32
template <class P>
LURK_HD F29<P> f29_const_r256() {
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = P::r(i);
    return f29_from_plain<P>(r);
}
// lazy Montgomery(2^261) value -> canonical Montgomery(2^256) 8 x 32 (the layout of the C ABI)
template <class P>
LURK_HD Fe<P> f29_to_mont256(const F29<P>& a) {
    F29<P> u = f29_mul<P>(f29_carry<P>(a), f29_const_r256<P>());  // a / 32: tight limbs, value < 2^255 + p < 4p
    uint32_t w[8];
    f29_pack<P>(u, w);
    fe_cond_sub2<P>(w);
    fe_cond_sub<P>(w);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
    return r;
}

}  // namespace lurk

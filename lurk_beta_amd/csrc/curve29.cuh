// curve29.cuh - XYZZ mixed addition on the radix-2^29 layer (field29.cuh): the inner loop of the MSM
// bucket accumulation.  Same group law and same exceptional cases as curve.cuh::xyzz_madd; only the
// representation of the accumulator differs.  Results leave through xyzz29_to_xyzz as ordinary
// Montgomery(2^256) XYZZ points, bit-compatible with the rest of the pipeline.
//
// Bound discipline (checked by tests/host_harness with LURK_F29_CHECK): the accumulator coordinates are
// "R-bounded" at every loop boundary: tight limbs (< 2^29, top limb < 2^27) and value < 2^259.
// Inside a mixed addition values grow through lazy subtractions (each adds the 64p bias, < 2^260); every
// such value is carried (limbs tight again) before it is multiplied, and the two stored sums X3, Y3 are
// brought back under 2^259 with f29_reduce (subtract floor(v / 2^254) * p: Pasta primes are 2^254 + eps).
#pragma once
#include "curve.cuh"
#include "field29.cuh"

#ifndef LURK_ACC_Y3_ROW
#define LURK_ACC_Y3_ROW 1
#endif
#ifndef LURK_ACC_AFFINE_FIRST
#define LURK_ACC_AFFINE_FIRST 1
#endif

namespace lurk {

// signed carry pass: limbs are int32 in (-2^31, 2^31); result tight, top limb keeps the rest (must be >= 0)
template <class P>
LURK_HD F29<P> f29_carry_signed(const int32_t* t) {
    F29<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t x = t[i] + c;
        r.l[i] = (uint32_t)x & F29_MASK;
        c = x >> 29;  // arithmetic shift
    }
    r.l[8] = (uint32_t)(t[8] + c);
    return r;
}

// v (tight limbs, top limb any u32 < 2^31) -> equivalent value < 2^255.1 with tight limbs.
// Pasta: p = 2^254 + eps, eps < 2^126 (limbs 0..4), so  v - k p = (v mod 2^254) - k eps  with k = v >> 254;
// adding p once keeps it positive:  (v mod 2^254) + 2^254 - (k - 1) eps.
template <class P>
LURK_HD F29<P> f29_reduce(const F29<P>& v) {
    static_assert(P::ID != 2, "f29_reduce is specialised for the Pasta primes");
    const uint32_t k = v.l[8] >> 22;                    // floor(v / 2^254)  (< 2^9)
    const uint32_t km1 = k ? k - 1 : 0;                 // k = 0: value is already < 2^254, add nothing, subtract nothing
    // e = (k-1) * eps as normalised 29-bit limbs (eps limbs = modulus limbs 0..4)
    int32_t t[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        c += (uint64_t)km1 * f29_mod<P>(i);
        t[i] = (int32_t)v.l[i] - (int32_t)((uint32_t)c & F29_MASK);
        c >>= 29;
    }
    t[5] = (int32_t)v.l[5] - (int32_t)(uint32_t)c;      // remaining carry of (k-1) eps (< 2^9)
    t[6] = (int32_t)v.l[6];
    t[7] = (int32_t)v.l[7];
    t[8] = (int32_t)(v.l[8] & ((1u << 22) - 1u)) + (k ? (int32_t)(1u << 22) : 0);
    return f29_carry_signed<P>(t);
}

template <class P>
struct Xyzz29 {
    F29<P> x, y, zz, zzz;
};

template <class P>
LURK_HD Xyzz<P> xyzz29_to_xyzz(const Xyzz29<P>& a, bool is_identity) {
    if (is_identity) return xyzz_identity<P>();
    Xyzz<P> r;
    r.x = f29_to_mont256<P>(a.x);
    r.y = f29_to_mont256<P>(a.y);
    r.zz = f29_to_mont256<P>(a.zz);
    r.zzz = f29_to_mont256<P>(a.zzz);
    return r;
}
template <class P>
LURK_HD void xyzz29_from_xyzz(const Xyzz<P>& a, Xyzz29<P>& r, bool& is_identity) {
    is_identity = xyzz_is_identity<P>(a);
    r.x = f29_from_mont256<P>(a.x);
    r.y = f29_from_mont256<P>(a.y);
    r.zz = f29_from_mont256<P>(a.zz);
    r.zzz = f29_from_mont256<P>(a.zzz);
}

// value == 0 mod p ?  for a carried value v < 2^261 (so v = k p with k < 128): exact, used after the cheap
// pre-filter v.l[0] < 128 (p == 1 mod 2^29, hence (k p) mod 2^29 == k)
template <class P>
LURK_HD bool f29_is_multiple_of_p(const F29<P>& v) {
    const uint32_t k = v.l[0];
    uint64_t c = 0;
    bool eq = true;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)k * f29_mod<P>(i);
        eq = eq && (v.l[i] == ((uint32_t)c & F29_MASK));
        c >>= 29;
    }
    c += (uint64_t)k * f29_mod<P>(8);
    return eq && (uint64_t)v.l[8] == c;
}

// exceptional case of the mixed addition: acc and (+/-)q share their x and y, the sum is 2(+/-q): affine
// doubling (a = 0 on both Pasta curves).  Rare, so out of line; arguments and result by value so that the
// caller's accumulator never has its address taken (it must stay in registers).
template <class P>
LURK_HD __attribute__((noinline)) Xyzz29<P> xyzz29_double_affine(F29<P> qx, F29<P> qy_signed) {
    const F29<P> u = f29_dbl<P>(qy_signed);                      // limbs < 2^30, value < 2^260
    const F29<P> v = f29_mul<P>(u, u);                           // < 2^259 + p
    const F29<P> w = f29_mul<P>(u, v);
    const F29<P> s = f29_mul<P>(qx, v);
    const F29<P> xx = f29_sqr<P>(qx);
    const F29<P> m = f29_carry<P>(f29_add<P>(f29_dbl<P>(xx), xx));  // 3 x^2, tight
    const F29<P> m2 = f29_sqr<P>(m);
    Xyzz29<P> r;
    r.x = f29_reduce<P>(f29_carry<P>(f29_sub<P>(m2, f29_dbl<P>(s))));
    const F29<P> t1 = f29_mul<P>(m, f29_sub<P>(s, r.x));
    const F29<P> t2 = f29_mul<P>(w, qy_signed);
    r.y = f29_reduce<P>(f29_carry<P>(f29_sub<P>(t1, t2)));
    r.zz = v;
    r.zzz = w;
    return r;
}

// acc += (+/-) q.  q is the 64-byte table record (Montgomery 2^256 limbs); acc_id tracks the identity.
// ACC_AFFINE: the caller knows zz = zzz = 1 (the accumulator holds exactly one base so far), which saves the four
// products with zz / zzz: 4 M + 2 S instead of 8 M + 2 S.  Every task's first real addition is of this kind, and the
// loop position is the same in all lanes of a wave, so the specialisation costs no divergence.
template <class P, bool ACC_AFFINE = false>
LURK_HD void xyzz29_madd(Xyzz29<P>& acc, bool& acc_id, const Affine<P>& q, bool negate) {
    if (affine_is_identity<P>(q)) return;
    const F29<P> qx = f29_from_mont256<P>(q.x);  // 32 x~ < 2^259, tight
    const F29<P> qy = f29_from_mont256<P>(q.y);
    if (acc_id) {
        acc.x = qx;
        acc.y = negate ? f29_reduce<P>(f29_carry<P>(f29_sub<P>(f29_zero<P>(), qy))) : qy;  // 64p - y, brought under 2^255.1
        Fe<P> one = fe_one<P>();
        acc.zz = f29_from_mont256<P>(one);
        acc.zzz = acc.zz;
        acc_id = false;
        return;
    }
    F29_ASSERT_LIMBS(acc.x, 29, "acc.x"); F29_ASSERT_TOP(acc.x, 27, "acc.x");
    F29_ASSERT_LIMBS(acc.y, 29, "acc.y"); F29_ASSERT_TOP(acc.y, 27, "acc.y");
    F29_ASSERT_LIMBS(acc.zz, 29, "acc.zz"); F29_ASSERT_TOP(acc.zz, 27, "acc.zz");
    F29_ASSERT_LIMBS(acc.zzz, 29, "acc.zzz"); F29_ASSERT_TOP(acc.zzz, 27, "acc.zzz");
    const F29<P> u2 = ACC_AFFINE ? qx : f29_mul<P>(qx, acc.zz);    // < 2^257.2  (affine: < 2^259)
    const F29<P> s2 = ACC_AFFINE ? qy : f29_mul<P>(qy, acc.zzz);
    const F29<P> p = f29_carry<P>(f29_sub<P>(u2, acc.x));  // U2 - X1 + 64p  < 2^260.3
    // r = S2 - Y1 for +q; for -q the true r is -(S2 + Y1): keep r' = S2 + Y1 and flip the sign of (Q - X3) below
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F29_ASSERT(f29_bias<P>(i) >= acc.y.l[i], "madd: Y1 limb above the bias");
        r.l[i] = s2.l[i] + (negate ? acc.y.l[i] : f29_bias<P>(i) - acc.y.l[i]);
    }
    r = f29_carry<P>(r);
    if (p.l[0] < 128u && f29_is_multiple_of_p<P>(p)) {
        if (r.l[0] < 128u && f29_is_multiple_of_p<P>(r)) {  // r (or r' = -r) == 0: equal points
            const F29<P> qys = negate ? f29_reduce<P>(f29_carry<P>(f29_sub<P>(f29_zero<P>(), qy))) : qy;
            acc = xyzz29_double_affine<P>(qx, qys);
        } else {
            acc_id = true;  // opposite points
        }
        return;
    }
    const F29<P> pp = f29_sqr<P>(p);          // < 2^259.7
    const F29<P> ppp = f29_mul<P>(p, pp);        // < 2^259.1
    const F29<P> qq = f29_mul<P>(acc.x, pp);     // < 2^257.8
    const F29<P> r2 = f29_sqr<P>(r);          // < 2^259.7
    // X3 = R^2 - PPP - 2Q   (two lazy subtractions: limbs < 2^32, value < 2^261.5)
    F29<P> x3 = f29_sub<P>(f29_sub<P>(r2, ppp), f29_dbl<P>(qq));
    x3 = f29_reduce<P>(f29_carry<P>(x3));        // < 2^255.1, tight
    // D = Q - X3 (or X3 - Q when r' = -r)
    F29<P> d;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F29_ASSERT(f29_bias<P>(i) >= qq.l[i] && f29_bias<P>(i) >= x3.l[i], "madd: Q / X3 limb above the bias");
        d.l[i] = negate ? x3.l[i] + (f29_bias<P>(i) - qq.l[i]) : qq.l[i] + (f29_bias<P>(i) - x3.l[i]);
    }
#if LURK_ACC_Y3_ROW
    // Y3 = R D - Y1 PPP as ONE lazy row: R * D + (64p - Y1) * PPP, a single Montgomery reduction
    // (operands tight: 18 products of < 2^58 per column); value < 2^259.7 + p
    Dot29<P> row;
    dot29_init<P>(row);
    dot29_mac<P>(row, r, f29_carry<P>(d));
    dot29_mac<P>(row, f29_carry<P>(f29_sub<P>(f29_zero<P>(), acc.y)), ppp);
    // (the f29_reduce is not optional: in the affine-accumulator form r and p reach 2^260.6, the row 2^521.4, so Y3 can exceed
    // 64p = 2^260 and the next addition's 64p - Y1 would go negative - seen on the GPU and reproduced on the host with points
    // 640..642 of the synthetic key; in the general form Y3 stays below 2^259.96, but dropping the reduce there measured no gain)
    F29<P> y3 = f29_reduce<P>(dot29_finish<P>(row));
#else
    const F29<P> t1 = f29_mul<P>(r, d);          // r tight (< 2^260.3), d loose (< 2^260.1): < 2^259.5
    const F29<P> t2 = f29_mul<P>(acc.y, ppp);    // < 2^257.2
    F29<P> y3 = f29_reduce<P>(f29_carry<P>(f29_sub<P>(t1, t2)));
#endif
    acc.x = x3;
    acc.y = y3;
    if (ACC_AFFINE) {
        acc.zz = f29_reduce<P>(pp);              // < 2^255.1 (the loop invariant wants < 2^259)
        acc.zzz = f29_reduce<P>(ppp);
    } else {
        acc.zz = f29_mul<P>(acc.zz, pp);         // < 2^257.8
        acc.zzz = f29_mul<P>(acc.zzz, ppp);      // < 2^257.2
    }
}

// Doubling of an R-bounded XYZZ29 point that is not the identity (dbl-2008-s-1, a = 0: 6 M + 3 S), result R-bounded: the link of
// the doubling chains that build a window table (msm_precompute.hip).  The bounds are xyzz29_double_affine's with the two
// products by zz / zzz of xyzz29_madd; a prime-order curve has no point with y = 0.
template <class P>
LURK_HD Xyzz29<P> xyzz29_dbl(const Xyzz29<P>& a) {
    const F29<P> u = f29_dbl<P>(a.y);                            // limbs < 2^30, value < 2^260
    const F29<P> v = f29_mul<P>(u, u);                           // < 2^259 + p
    const F29<P> w = f29_mul<P>(u, v);
    const F29<P> s = f29_mul<P>(a.x, v);
    const F29<P> xx = f29_sqr<P>(a.x);
    const F29<P> m = f29_carry<P>(f29_add<P>(f29_dbl<P>(xx), xx));  // 3 x^2, tight
    const F29<P> m2 = f29_sqr<P>(m);
    Xyzz29<P> r;
    r.x = f29_reduce<P>(f29_carry<P>(f29_sub<P>(m2, f29_dbl<P>(s))));
    const F29<P> t1 = f29_mul<P>(m, f29_sub<P>(s, r.x));
    const F29<P> t2 = f29_mul<P>(w, a.y);
    r.y = f29_reduce<P>(f29_carry<P>(f29_sub<P>(t1, t2)));
    r.zz = f29_mul<P>(v, a.zz);
    r.zzz = f29_mul<P>(w, a.zzz);
    return r;
}

// Doubling of a general XYZZ29 point: only reached when two equal partial sums meet in a reduction tree - rare, so it goes
// through the 32-bit-limb group law (out of line; arguments and result by value, see xyzz29_double_affine).
template <class P>
LURK_HD __attribute__((noinline)) Xyzz29<P> xyzz29_double_general(Xyzz29<P> a) {
    const Xyzz<P> d = xyzz_dbl<P>(xyzz29_to_xyzz<P>(a, false));
    Xyzz29<P> r;
    bool id;
    xyzz29_from_xyzz<P>(d, r, id);  // a doubled point of a prime-order curve is never the identity
    return r;
}

// acc += q, both on the radix-2^29 layer (add-2008-s: 12 M + 2 S), identities tracked by the flags.  The nodes of the reduction
// trees of the small-commitment path (msm_small.hip).  Bounds as in xyzz29_madd: both inputs R-bounded (tight limbs, value < 2^259),
// the result R-bounded again; p and r are carried differences (< 2^260.3), X3 and Y3 leave through f29_reduce.
template <class P>
LURK_HD void xyzz29_add(Xyzz29<P>& acc, bool& acc_id, const Xyzz29<P>& q, bool q_id) {
    if (q_id) return;
    if (acc_id) {
        acc = q;
        acc_id = false;
        return;
    }
    F29_ASSERT_LIMBS(acc.x, 29, "add acc.x"); F29_ASSERT_TOP(acc.x, 27, "add acc.x");
    F29_ASSERT_LIMBS(acc.y, 29, "add acc.y"); F29_ASSERT_TOP(acc.y, 27, "add acc.y");
    F29_ASSERT_LIMBS(acc.zz, 29, "add acc.zz"); F29_ASSERT_TOP(acc.zz, 27, "add acc.zz");
    F29_ASSERT_LIMBS(acc.zzz, 29, "add acc.zzz"); F29_ASSERT_TOP(acc.zzz, 27, "add acc.zzz");
    F29_ASSERT_LIMBS(q.x, 29, "add q.x"); F29_ASSERT_TOP(q.x, 27, "add q.x");
    F29_ASSERT_LIMBS(q.y, 29, "add q.y"); F29_ASSERT_TOP(q.y, 27, "add q.y");
    F29_ASSERT_LIMBS(q.zz, 29, "add q.zz"); F29_ASSERT_TOP(q.zz, 27, "add q.zz");
    F29_ASSERT_LIMBS(q.zzz, 29, "add q.zzz"); F29_ASSERT_TOP(q.zzz, 27, "add q.zzz");
    const F29<P> u1 = f29_mul<P>(acc.x, q.zz);     // < 2^257.2, tight
    const F29<P> u2 = f29_mul<P>(q.x, acc.zz);
    const F29<P> s1 = f29_mul<P>(acc.y, q.zzz);
    const F29<P> s2 = f29_mul<P>(q.y, acc.zzz);
    const F29<P> p = f29_carry<P>(f29_sub<P>(u2, u1));  // U2 - U1 + 64p < 2^260.3
    const F29<P> r = f29_carry<P>(f29_sub<P>(s2, s1));
    if (p.l[0] < 128u && f29_is_multiple_of_p<P>(p)) {
        if (r.l[0] < 128u && f29_is_multiple_of_p<P>(r)) acc = xyzz29_double_general<P>(acc);  // equal points
        else acc_id = true;                                                                     // opposite points
        return;
    }
    const F29<P> pp = f29_sqr<P>(p);               // < 2^259.7
    const F29<P> ppp = f29_mul<P>(p, pp);          // < 2^259.1
    const F29<P> qq = f29_mul<P>(u1, pp);          // < 2^256
    const F29<P> r2 = f29_sqr<P>(r);               // < 2^259.7
    F29<P> x3 = f29_sub<P>(f29_sub<P>(r2, ppp), f29_dbl<P>(qq));
    x3 = f29_reduce<P>(f29_carry<P>(x3));          // < 2^255.1, tight
    F29<P> d;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        F29_ASSERT(f29_bias<P>(i) >= x3.l[i], "add: X3 limb above the bias");
        d.l[i] = qq.l[i] + (f29_bias<P>(i) - x3.l[i]);
    }
    // Y3 = R (Q - X3) - S1 PPP as one lazy row (see xyzz29_madd): value < 2^259.9 + p before the reduce
    Dot29<P> row;
    dot29_init<P>(row);
    dot29_mac<P>(row, r, f29_carry<P>(d));
    dot29_mac<P>(row, f29_carry<P>(f29_sub<P>(f29_zero<P>(), s1)), ppp);
    const F29<P> y3 = f29_reduce<P>(dot29_finish<P>(row));
    const F29<P> zz12 = f29_mul<P>(acc.zz, q.zz);  // < 2^257.2
    const F29<P> zzz12 = f29_mul<P>(acc.zzz, q.zzz);
    acc.x = x3;
    acc.y = y3;
    acc.zz = f29_mul<P>(zz12, pp);                 // < 2^256
    acc.zzz = f29_mul<P>(zzz12, ppp);
}

// Sum of nt XYZZ points held in the 8 x 32 form (a bucket's task partials: msm_finalize.hip).  One point is copied, more go through
// the radix-2^29 general addition (6.4 us per dependent addition on a lane where the 8 x 32 group law takes ~20).
template <class P>
LURK_HD Xyzz<P> xyzz_sum_via29(const Xyzz<P>* pts, uint32_t nt) {
    if (nt == 0) return xyzz_identity<P>();
    if (nt == 1) return pts[0];
    Xyzz29<P> acc, q;
    bool acc_id, q_id;
    xyzz29_from_xyzz<P>(pts[0], acc, acc_id);
    for (uint32_t i = 1; i < nt; i++) {
        xyzz29_from_xyzz<P>(pts[i], q, q_id);
        xyzz29_add<P>(acc, acc_id, q, q_id);
    }
    return xyzz29_to_xyzz<P>(acc, acc_id);
}

// One accumulation task on the radix-2^29 layer: the signed table records sorted[first .. last) summed into (acc, acc_id), which
// leave R-bounded (msm_task_accumulate29 returns them as an ordinary XYZZ point; msm_bucket_direct.hip keeps them on this layer).
template <class P>
LURK_HD void msm_task_accumulate29_raw(const uint32_t* sorted, uint32_t first, uint32_t last, const Affine<P>* table, Xyzz29<P>& acc, bool& acc_id) {
    acc.x = acc.y = acc.zz = acc.zzz = f29_zero<P>();
    acc_id = true;
    // (gathering the next base ahead of the current addition was measured: no gain, the other waves of the SIMD
    // already cover the load; so was requesting one dword of it ahead - a one-VGPR "touch" for L2 and the TLB - 3.6 against
    // 3.5 ms, although confining every gather to a 64 MB window of the 3.25 GiB table does make the kernel 8 % faster)
#if LURK_ACC_AFFINE_FIRST
    uint32_t j = first;
    if (j < last) {  // first base: a copy (zz = zzz = 1)
        const uint32_t e = sorted[j++];
        xyzz29_madd<P>(acc, acc_id, table[e & 0x7fffffffu], (e & 0x80000000u) != 0);
    }
    if (j < last) {  // second base: affine + affine (unless the first was the identity record)
        const uint32_t e = sorted[j++];
        const Affine<P> q = table[e & 0x7fffffffu];
#if LURK_ACC_AFFINE_FIRST
        if (!acc_id) xyzz29_madd<P, true>(acc, acc_id, q, (e & 0x80000000u) != 0);
        else
#endif
            xyzz29_madd<P>(acc, acc_id, q, (e & 0x80000000u) != 0);
    }
    for (; j < last; j++) {
        const uint32_t e = sorted[j];
        const Affine<P> q = table[e & 0x7fffffffu];
        xyzz29_madd<P>(acc, acc_id, q, (e & 0x80000000u) != 0);
    }
#else
    for (uint32_t j = first; j < last; j++) {
        const uint32_t e = sorted[j];
#ifdef LURK_ACC_DEBUG_TABLE_MASK  // timing experiment only (wrong results): every gather lands in a small window of the table
        const Affine<P> q = table[e & LURK_ACC_DEBUG_TABLE_MASK];
#else
        const Affine<P> q = table[e & 0x7fffffffu];
#endif
        xyzz29_madd<P>(acc, acc_id, q, (e & 0x80000000u) != 0);
    }
#endif
}
template <class P>
LURK_HD Xyzz<P> msm_task_accumulate29(const uint32_t* sorted, uint32_t first, uint32_t last, const Affine<P>* table) {
    Xyzz29<P> acc;
    bool acc_id;
    msm_task_accumulate29_raw<P>(sorted, first, last, table, acc, acc_id);
    return xyzz29_to_xyzz<P>(acc, acc_id);
}

}  // namespace lurk

// curve.cuh - short-Weierstrass arithmetic for the Pasta curves (y^2 = x^3 + 5, a = 0) in XYZZ
// coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity: ZZ = 0).  Shared host/device.
//
// XYZZ is what the bucket accumulation of the Pippenger MSM runs on: a mixed addition
// (accumulator += affine base) is 8M + 2S with no inversion, against 7M + 4S for Jacobian.
// The C ABI carries the pasta_curves `repr-c` layouts (affine {x,y}, Jacobian {x,y,z},
// Montgomery limbs); conversions live here too.
#pragma once
#include "field.cuh"

namespace lurk {

template <class P>
struct Affine {  // identity = (0, 0)
    Fe<P> x, y;
};
template <class P>
struct Xyzz {
    Fe<P> x, y, zz, zzz;
};
template <class P>
struct Jacobian {  // identity: z = 0
    Fe<P> x, y, z;
};

template <class P>
LURK_HD bool affine_is_identity(const Affine<P>& a) {
    return fe_is_zero<P>(a.x) && fe_is_zero<P>(a.y);
}
template <class P>
LURK_HD Xyzz<P> xyzz_identity() {
    Xyzz<P> r;
    r.x = fe_zero<P>();
    r.y = fe_zero<P>();
    r.zz = fe_zero<P>();
    r.zzz = fe_zero<P>();
    return r;
}
template <class P>
LURK_HD bool xyzz_is_identity(const Xyzz<P>& p) {
    return fe_is_zero<P>(p.zz);
}
template <class P>
LURK_HD Xyzz<P> xyzz_from_affine(const Affine<P>& a) {
    if (affine_is_identity<P>(a)) return xyzz_identity<P>();
    Xyzz<P> r;
    r.x = a.x;
    r.y = a.y;
    r.zz = fe_one<P>();
    r.zzz = fe_one<P>();
    return r;
}

// 2*(x, y) for an affine point (dbl-2008-s-1 with ZZ = 1, a = 0)
template <class P>
LURK_HD Xyzz<P> xyzz_dbl_affine(const Fe<P>& x1, const Fe<P>& y1) {
    if (fe_is_zero<P>(y1)) return xyzz_identity<P>();  // order-2 point (none on Pasta; kept for safety)
    Fe<P> u = fe_dbl<P>(y1);
    Fe<P> v = fe_sqr<P>(u);
    Fe<P> w = fe_mul<P>(u, v);
    Fe<P> s = fe_mul<P>(x1, v);
    Fe<P> xx = fe_sqr<P>(x1);
    Fe<P> m = fe_add<P>(fe_dbl<P>(xx), xx);
    Xyzz<P> r;
    r.x = fe_sub<P>(fe_sqr<P>(m), fe_dbl<P>(s));
    r.y = fe_sub<P>(fe_mul<P>(m, fe_sub<P>(s, r.x)), fe_mul<P>(w, y1));
    r.zz = v;
    r.zzz = w;
    return r;
}

// general doubling (dbl-2008-s-1, a = 0)
template <class P>
LURK_HD Xyzz<P> xyzz_dbl(const Xyzz<P>& p) {
    if (xyzz_is_identity<P>(p) || fe_is_zero<P>(p.y)) return xyzz_identity<P>();
    Fe<P> u = fe_dbl<P>(p.y);
    Fe<P> v = fe_sqr<P>(u);
    Fe<P> w = fe_mul<P>(u, v);
    Fe<P> s = fe_mul<P>(p.x, v);
    Fe<P> xx = fe_sqr<P>(p.x);
    Fe<P> m = fe_add<P>(fe_dbl<P>(xx), xx);
    Xyzz<P> r;
    r.x = fe_sub<P>(fe_sqr<P>(m), fe_dbl<P>(s));
    r.y = fe_sub<P>(fe_mul<P>(m, fe_sub<P>(s, r.x)), fe_mul<P>(w, p.y));
    r.zz = fe_mul<P>(v, p.zz);
    r.zzz = fe_mul<P>(w, p.zzz);
    return r;
}

// acc += (+/-) affine point (madd-2008-s), with every exceptional case handled
template <class P>
LURK_HD void xyzz_madd(Xyzz<P>& acc, const Affine<P>& q, bool negate) {
    if (affine_is_identity<P>(q)) return;
    Fe<P> qy = negate ? fe_neg<P>(q.y) : q.y;
    if (xyzz_is_identity<P>(acc)) {
        acc.x = q.x;
        acc.y = qy;
        acc.zz = fe_one<P>();
        acc.zzz = fe_one<P>();
        return;
    }
    Fe<P> u2 = fe_mul<P>(q.x, acc.zz);
    Fe<P> s2 = fe_mul<P>(qy, acc.zzz);
    Fe<P> p = fe_sub<P>(u2, acc.x);
    Fe<P> r = fe_sub<P>(s2, acc.y);
    if (fe_is_zero<P>(p)) {
        if (fe_is_zero<P>(r)) acc = xyzz_dbl_affine<P>(q.x, qy);
        else acc = xyzz_identity<P>();
        return;
    }
    Fe<P> pp = fe_sqr<P>(p);
    Fe<P> ppp = fe_mul<P>(p, pp);
    Fe<P> qq = fe_mul<P>(acc.x, pp);
    Fe<P> x3 = fe_sub<P>(fe_sub<P>(fe_sqr<P>(r), ppp), fe_dbl<P>(qq));
    Fe<P> y3 = fe_sub<P>(fe_mul<P>(r, fe_sub<P>(qq, x3)), fe_mul<P>(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul<P>(acc.zz, pp);
    acc.zzz = fe_mul<P>(acc.zzz, ppp);
}

// acc += q (add-2008-s), with every exceptional case handled
template <class P>
LURK_HD void xyzz_add(Xyzz<P>& acc, const Xyzz<P>& q) {
    if (xyzz_is_identity<P>(q)) return;
    if (xyzz_is_identity<P>(acc)) {
        acc = q;
        return;
    }
    Fe<P> u1 = fe_mul<P>(acc.x, q.zz);
    Fe<P> u2 = fe_mul<P>(q.x, acc.zz);
    Fe<P> s1 = fe_mul<P>(acc.y, q.zzz);
    Fe<P> s2 = fe_mul<P>(q.y, acc.zzz);
    Fe<P> p = fe_sub<P>(u2, u1);
    Fe<P> r = fe_sub<P>(s2, s1);
    if (fe_is_zero<P>(p)) {
        if (fe_is_zero<P>(r)) acc = xyzz_dbl<P>(acc);
        else acc = xyzz_identity<P>();
        return;
    }
    Fe<P> pp = fe_sqr<P>(p);
    Fe<P> ppp = fe_mul<P>(p, pp);
    Fe<P> qq = fe_mul<P>(u1, pp);
    Fe<P> x3 = fe_sub<P>(fe_sub<P>(fe_sqr<P>(r), ppp), fe_dbl<P>(qq));
    Fe<P> y3 = fe_sub<P>(fe_mul<P>(r, fe_sub<P>(qq, x3)), fe_mul<P>(s1, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul<P>(fe_mul<P>(acc.zz, q.zz), pp);
    acc.zzz = fe_mul<P>(fe_mul<P>(acc.zzz, q.zzz), ppp);
}

// k * p for a small non-negative integer k (double-and-add, msb first)
template <class P>
LURK_HD Xyzz<P> xyzz_mul_small(const Xyzz<P>& p, uint32_t k) {
    Xyzz<P> acc = xyzz_identity<P>();
    int top = 31;
    while (top >= 0 && !((k >> top) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        acc = xyzz_dbl<P>(acc);
        if ((k >> i) & 1) xyzz_add<P>(acc, p);
    }
    return acc;
}

template <class P>
LURK_HD Affine<P> xyzz_to_affine(const Xyzz<P>& p) {
    Affine<P> a;
    if (xyzz_is_identity<P>(p)) {
        a.x = fe_zero<P>();
        a.y = fe_zero<P>();
        return a;
    }
    // 1/ZZZ gives both: 1/ZZ = ZZ^2 / ZZZ^2 ... simpler: two uses of one inversion of ZZ*ZZZ
    Fe<P> t = fe_inv<P>(fe_mul<P>(p.zz, p.zzz));
    Fe<P> izz = fe_mul<P>(t, p.zzz);
    Fe<P> izzz = fe_mul<P>(t, p.zz);
    a.x = fe_mul<P>(p.x, izz);
    a.y = fe_mul<P>(p.y, izzz);
    return a;
}
// two points with ONE field inversion (Montgomery's trick over ZZ ZZZ of both); the same affine coordinates as xyzz_to_affine
template <class P>
LURK_HD void xyzz_pair_to_affine(const Xyzz<P>& p, const Xyzz<P>& q, Affine<P>& ap, Affine<P>& aq) {
    if (xyzz_is_identity<P>(p) || xyzz_is_identity<P>(q)) {
        ap = xyzz_to_affine<P>(p);
        aq = xyzz_to_affine<P>(q);
        return;
    }
    const Fe<P> dp = fe_mul<P>(p.zz, p.zzz), dq = fe_mul<P>(q.zz, q.zzz);
    const Fe<P> t = fe_inv<P>(fe_mul<P>(dp, dq));
    const Fe<P> tp = fe_mul<P>(t, dq), tq = fe_mul<P>(t, dp);  // 1 / (ZZ ZZZ) of p, of q
    ap.x = fe_mul<P>(p.x, fe_mul<P>(tp, p.zzz));
    ap.y = fe_mul<P>(p.y, fe_mul<P>(tp, p.zz));
    aq.x = fe_mul<P>(q.x, fe_mul<P>(tq, q.zzz));
    aq.y = fe_mul<P>(q.y, fe_mul<P>(tq, q.zz));
}
template <class P>
LURK_HD Xyzz<P> xyzz_from_jacobian(const Jacobian<P>& j) {
    if (fe_is_zero<P>(j.z)) return xyzz_identity<P>();
    Xyzz<P> r;
    r.x = j.x;
    r.y = j.y;
    r.zz = fe_sqr<P>(j.z);
    r.zzz = fe_mul<P>(r.zz, j.z);
    return r;
}
// Jacobian representative with Z = 1 (or the identity, z = 0): what the C ABI returns
template <class P>
LURK_HD Jacobian<P> jacobian_from_affine(const Affine<P>& a) {
    Jacobian<P> j;
    if (affine_is_identity<P>(a)) {
        j.x = fe_zero<P>();
        j.y = fe_zero<P>();
        j.z = fe_zero<P>();
        return j;
    }
    j.x = a.x;
    j.y = a.y;
    j.z = fe_one<P>();
    return j;
}

}  // namespace lurk

// Host-side harness: compiles the library's host/device headers (field.cuh, curve.cuh, ...) with
// plain g++ so their arithmetic can be checked against the oracle without a GPU.  TEST ONLY: the
// product never runs these on the CPU; the kernels in lurk_beta_amd/csrc/*.hip are the product.
// every limb / accumulator bound the radix-2^29 layer relies on is asserted at run time in this host build: the switch has to
// precede the FIRST inclusion of field29.cuh (poseidon29.cuh pulls it in)
#define LURK_F29_CHECK 1
#include <stddef.h>
#include <vector>
#include "../../lurk_beta_amd/csrc/field.cuh"
using namespace lurk;

template <class P>
static void mul_n(const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n, int op) {
    for (size_t i = 0; i < n; i++) {
        Fe<P> x, y, r;
        for (int k = 0; k < 8; k++) { x.l[k] = a[8 * i + k]; y.l[k] = b[8 * i + k]; }
        switch (op) {
            case 0: r = fe_mul<P>(x, y); break;
            case 1: r = fe_add<P>(x, y); break;
            case 2: r = fe_sub<P>(x, y); break;
            case 3: r = fe_sqr<P>(x); break;
            case 4: r = fe_inv<P>(x); break;
            case 5: r = fe_to_mont<P>(x); break;
            case 6: r = fe_from_mont<P>(x); break;
            case 7: r = fe_neg<P>(x); break;
            case 8: r = fe_mul_fips<P>(x, y); break;   // device column-wise schedule, portable primitives
            case 9: r = fe_mul_cios<P>(x, y); break;
            default: r = fe_zero<P>();
        }
        for (int k = 0; k < 8; k++) o[8 * i + k] = r.l[k];
    }
}
// inner product with one reduction: sum_i a[i]*b[i] over T operands (T = 3, 5, 9)
template <class P, int T>
static void dot_t(const uint32_t* a, const uint32_t* b, uint32_t* o) {
    DotAcc<P> A;
    dot_init<P>(A);
    for (int i = 0; i < T; i++) {
        Fe<P> x, y;
        for (int k = 0; k < 8; k++) { x.l[k] = a[8 * i + k]; y.l[k] = b[8 * i + k]; }
        dot_mac<P>(A, x, y);
    }
    Fe<P> r = dot_finish<P, T>(A);
    for (int k = 0; k < 8; k++) o[k] = r.l[k];
}
extern "C" void hh_fe_dot(int field, int T, const uint32_t* a, const uint32_t* b, uint32_t* o) {
#define DOT_CASE(P)                                   \
    if (T == 3) dot_t<P, 3>(a, b, o);                 \
    else if (T == 5) dot_t<P, 5>(a, b, o);            \
    else dot_t<P, 9>(a, b, o);
    if (field == 0) { DOT_CASE(PallasFp) } else if (field == 1) { DOT_CASE(PallasFq) } else { DOT_CASE(Bn254Fr) }
}
// the host's inverse (binary extended Euclid, field.cuh: host_inv) against the exponentiation the device uses; n Montgomery elements
template <class F>
static void fe_inv_both(const uint32_t* a, uint32_t* o_host, uint32_t* o_pow, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fe<F> x;
        for (int k = 0; k < 8; k++) x.l[k] = a[8 * i + k];
        const Fe<F> h = fe_inv<F>(x), q = fe_inv_pow<F>(x);
        for (int k = 0; k < 8; k++) { o_host[8 * i + k] = h.l[k]; o_pow[8 * i + k] = q.l[k]; }
    }
}
extern "C" void hh_fe_inv_both(int field, const uint32_t* a, uint32_t* o_host, uint32_t* o_pow, size_t n) {
    if (field == 0) fe_inv_both<PallasFp>(a, o_host, o_pow, n);
    else if (field == 1) fe_inv_both<PallasFq>(a, o_host, o_pow, n);
    else fe_inv_both<Bn254Fr>(a, o_host, o_pow, n);
}

extern "C" void hh_fe_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
    if (field == 0) mul_n<PallasFp>(a, b, o, n, op);
    else if (field == 1) mul_n<PallasFq>(a, b, o, n, op);
    else mul_n<Bn254Fr>(a, b, o, n, op);
}

// ---------------------------------------------------------------------------------------------
#include "../../lurk_beta_amd/csrc/poseidon.cuh"
#include "../../lurk_beta_amd/csrc/poseidon_params.hpp"
#include "../../lurk_beta_amd/csrc/poseidon29.cuh"

template <class P, int T>
static void poseidon_n(int mode, const uint32_t* pre, size_t n, uint32_t* out) {
    PoseidonParams<P> pp = make_poseidon_params<P>(T - 1);
    std::vector<uint32_t> img = poseidon_device_image<P>(pp);
    if (mode == 2) {  // the radix-2^29 permutation the kernels run
        std::vector<uint32_t> img29 = poseidon29_image<P>(img);
        const PoseidonLayout<T> L(pp.rf, pp.rp);
        for (size_t h = 0; h < n; h++) {
            F29<P> s[T];
            s[0] = ld_const29<P>(img29.data());
            for (int i = 1; i < T; i++) s[i] = poseidon29_from_canonical<P>(pre + (h * (T - 1) + (i - 1)) * 8, img29.data() + (size_t)L.total() * P29_STRIDE);
            poseidon29_permute<P, T>(s, img29.data(), pp.rf, pp.rp);
            Fe<P> d = poseidon29_to_canonical<P>(s[1]);
            for (int k = 0; k < 8; k++) out[h * 8 + k] = d.l[k];
        }
        return;
    }
    for (size_t h = 0; h < n; h++) {
        Fe<P> s[T];
        s[0] = pp.domain_tag;
        for (int i = 1; i < T; i++) {
            Fe<P> x;
            for (int k = 0; k < 8; k++) x.l[k] = pre[(h * (T - 1) + (i - 1)) * 8 + k];
            s[i] = fe_to_mont<P>(x);
        }
        if (mode == 0) poseidon_permute<P, T>(s, (const Fe<P>*)img.data(), pp.rf, pp.rp);
        else poseidon_permute_plain<P, T>(s, pp.rc.data(), pp.mds.data(), pp.rf, pp.rp);
        Fe<P> d = fe_from_mont<P>(s[1]);
        for (int k = 0; k < 8; k++) out[h * 8 + k] = d.l[k];
    }
}
template <class P>
static void poseidon_f(int arity, int mode, const uint32_t* pre, size_t n, uint32_t* out) {
    switch (arity) {
        case 3: poseidon_n<P, 4>(mode, pre, n, out); break;
        case 4: poseidon_n<P, 5>(mode, pre, n, out); break;
        case 6: poseidon_n<P, 7>(mode, pre, n, out); break;
        case 8: poseidon_n<P, 9>(mode, pre, n, out); break;
    }
}
extern "C" void hh_poseidon(int field, int arity, int mode, const uint32_t* pre, size_t n, uint32_t* out) {
    if (field == 0) poseidon_f<PallasFp>(arity, mode, pre, n, out);
    else if (field == 1) poseidon_f<PallasFq>(arity, mode, pre, n, out);
    else poseidon_f<Bn254Fr>(arity, mode, pre, n, out);
}
// Slot-witness trace of one Poseidon hash on the radix-2^29 layer (poseidon29_permute_trace): per hash
// [preimage | (l^2, l^4, l^5 + key) per S-box in circuit order | digest], canonical values, (arity + 3 * sboxes + 1) x 8 words
template <class P, int T>
static void poseidon_trace_n(const uint32_t* pre, size_t n, uint32_t* out) {
    PoseidonParams<P> pp = make_poseidon_params<P>(T - 1);
    std::vector<Fe<P>> post = neptune_post_keys<P>(pp);
    std::vector<uint32_t> post_words;
    for (auto& x : post) for (int k = 0; k < 8; k++) post_words.push_back(x.l[k]);
    std::vector<uint32_t> img29 = poseidon29_image<P>(poseidon_device_image<P>(pp), post_words);
    const PoseidonLayout<T> L(pp.rf, pp.rp);
    const uint32_t* mont2 = img29.data() + (size_t)L.total() * P29_STRIDE;
    const uint32_t* post29 = mont2 + P29_STRIDE;
    const size_t nsbox = (size_t)T * pp.rf + pp.rp, per = (T - 1) + 3 * nsbox + 1;
    for (size_t h = 0; h < n; h++) {
        uint32_t* o = out + h * per * 8;
        auto put = [&](size_t idx, const F29<P>& v) {
            Fe<P> d = poseidon29_to_canonical<P>(v);
            for (int k = 0; k < 8; k++) o[idx * 8 + k] = d.l[k];
        };
        F29<P> s[T];
        s[0] = ld_const29<P>(img29.data());
        for (int i = 1; i < T; i++) {
            s[i] = poseidon29_from_canonical<P>(pre + (h * (T - 1) + (i - 1)) * 8, mont2);
            put(i - 1, s[i]);
        }
        auto emit = [&](int sbox, const F29<P>& l2, const F29<P>& l4, const F29<P>& l5k) {
            put((T - 1) + 3 * (size_t)sbox, l2);
            put((T - 1) + 3 * (size_t)sbox + 1, l4);
            put((T - 1) + 3 * (size_t)sbox + 2, l5k);
        };
        poseidon29_permute_trace<P, T>(s, img29.data(), post29, pp.rf, pp.rp, emit);
        put(per - 1, s[1]);
    }
}
template <class P>
static void poseidon_trace_f(int arity, const uint32_t* pre, size_t n, uint32_t* out) {
    switch (arity) {
        case 3: poseidon_trace_n<P, 4>(pre, n, out); break;
        case 4: poseidon_trace_n<P, 5>(pre, n, out); break;
        case 6: poseidon_trace_n<P, 7>(pre, n, out); break;
        case 8: poseidon_trace_n<P, 9>(pre, n, out); break;
    }
}
extern "C" void hh_poseidon_trace(int field, int arity, const uint32_t* pre, size_t n, uint32_t* out) {
    if (field == 0) poseidon_trace_f<PallasFp>(arity, pre, n, out);
    else if (field == 1) poseidon_trace_f<PallasFq>(arity, pre, n, out);
    else poseidon_trace_f<Bn254Fr>(arity, pre, n, out);
}
// canonical round constants + mds + (rf, rp) for comparison with the oracle's generator
template <class P>
static int params_f(int arity, int* rf, int* rp, uint32_t* rc, uint32_t* mds) {
    PoseidonParams<P> pp = make_poseidon_params<P>(arity);
    *rf = pp.rf; *rp = pp.rp;
    for (size_t i = 0; i < pp.rc.size(); i++) { Fe<P> c = fe_from_mont<P>(pp.rc[i]); for (int k = 0; k < 8; k++) rc[i * 8 + k] = c.l[k]; }
    for (size_t i = 0; i < pp.mds.size(); i++) { Fe<P> c = fe_from_mont<P>(pp.mds[i]); for (int k = 0; k < 8; k++) mds[i * 8 + k] = c.l[k]; }
    return (int)pp.rc.size();
}
extern "C" int hh_poseidon_params(int field, int arity, int* rf, int* rp, uint32_t* rc, uint32_t* mds) {
    if (field == 0) return params_f<PallasFp>(arity, rf, rp, rc, mds);
    if (field == 1) return params_f<PallasFq>(arity, rf, rp, rc, mds);
    return params_f<Bn254Fr>(arity, rf, rp, rc, mds);
}

// ---------------------------------------------------------------------------------------------
#include "../../lurk_beta_amd/csrc/curve29.cuh"

// mode 0: acc = sum (+/-) P_i with xyzz_madd; mode 1: pairwise xyzz_add of xyzz_from_affine;
// mode 2: sum k_i * P_i with xyzz_mul_small (k_i = signs[i] as small integer); mode 3: as mode 0 on the
// radix-2^29 layer (xyzz29_madd, bound assertions enabled; the second base goes through the affine-accumulator
// specialisation exactly as in msm_task_accumulate29); mode 4: radix-2^29, general addition only; mode 5: partial sums of three
// bases each (8 x 32) summed by xyzz_sum_via29 (msm_finalize.hip).  Output: affine Montgomery.
template <class P>
static void curve_sum(int mode, const uint32_t* bases, const uint32_t* signs, size_t n, uint32_t* out) {
    Xyzz<P> acc = xyzz_identity<P>();
    Xyzz29<P> acc29;
    acc29.x = acc29.y = acc29.zz = acc29.zzz = f29_zero<P>();
    bool acc29_id = true;
    if (mode == 5) {  // the finalize stage: task partials (8 x 32 XYZZ sums of 3 bases each) summed by xyzz_sum_via29
        std::vector<Xyzz<P>> partials;
        for (size_t i = 0; i < n; i++) {
            Affine<P> a;
            for (int k = 0; k < 8; k++) { a.x.l[k] = bases[i * 16 + k]; a.y.l[k] = bases[i * 16 + 8 + k]; }
            if (i % 3 == 0) partials.push_back(xyzz_identity<P>());
            xyzz_madd<P>(partials.back(), a, signs[i] != 0);
        }
        acc = xyzz_sum_via29<P>(partials.data(), (uint32_t)partials.size());
        n = 0;
    }
    for (size_t i = 0; i < n; i++) {
        Affine<P> a;
        for (int k = 0; k < 8; k++) { a.x.l[k] = bases[i * 16 + k]; a.y.l[k] = bases[i * 16 + 8 + k]; }
        if (mode == 3 && i == 1 && !acc29_id) xyzz29_madd<P, true>(acc29, acc29_id, a, signs[i] != 0);  // the kernel's schedule
        else if (mode == 3 || mode == 4) xyzz29_madd<P>(acc29, acc29_id, a, signs[i] != 0);
        else if (mode == 0) xyzz_madd<P>(acc, a, signs[i] != 0);
        else if (mode == 1) {
            Xyzz<P> q = xyzz_from_affine<P>(a);
            if (signs[i]) q.y = fe_neg<P>(q.y);
            xyzz_add<P>(acc, q);
        } else {
            Xyzz<P> q = xyzz_mul_small<P>(xyzz_from_affine<P>(a), signs[i]);
            xyzz_add<P>(acc, q);
        }
    }
    if (mode == 3 || mode == 4) acc = xyzz29_to_xyzz<P>(acc29, acc29_id);
    Affine<P> r = xyzz_to_affine<P>(acc);
    // round-trip through the Jacobian helpers as well
    Jacobian<P> j = jacobian_from_affine<P>(r);
    r = xyzz_to_affine<P>(xyzz_from_jacobian<P>(j));
    for (int k = 0; k < 8; k++) { out[k] = r.x.l[k]; out[8 + k] = r.y.l[k]; }
}
// The reduction tree of the small-commitment path (msm_small.hip) on the host with the bound assertions on: the n signed bases are
// dealt round-robin to `lanes` accumulators (xyzz29_madd), which are then summed by an xor butterfly of xyzz29_add - every "lane"
// computes its own copy of every level, as the wave does.  Output: affine Montgomery.
template <class P>
static void curve_tree(const uint32_t* bases, const uint32_t* signs, size_t n, int lanes, uint32_t* out) {
    std::vector<Xyzz29<P>> acc(lanes);
    std::vector<char> id(lanes, 1);
    for (int l = 0; l < lanes; l++) acc[l].x = acc[l].y = acc[l].zz = acc[l].zzz = f29_zero<P>();
    for (size_t i = 0; i < n; i++) {
        Affine<P> a;
        for (int k = 0; k < 8; k++) { a.x.l[k] = bases[i * 16 + k]; a.y.l[k] = bases[i * 16 + 8 + k]; }
        bool b = id[i % lanes] != 0;
        xyzz29_madd<P>(acc[i % lanes], b, a, signs[i] != 0);
        id[i % lanes] = b;
    }
    for (int off = 1; off < lanes; off <<= 1) {
        std::vector<Xyzz29<P>> nxt = acc;
        std::vector<char> nid = id;
        for (int l = 0; l < lanes; l++) {
            bool b = id[l] != 0;
            xyzz29_add<P>(nxt[l], b, acc[l ^ off], id[l ^ off] != 0);
            nid[l] = b;
        }
        acc.swap(nxt);
        id.swap(nid);
    }
    Affine<P> r = xyzz_to_affine<P>(xyzz29_to_xyzz<P>(acc[0], id[0] != 0));
    for (int l = 1; l < lanes; l++) {  // every lane must hold the same total
        Affine<P> o = xyzz_to_affine<P>(xyzz29_to_xyzz<P>(acc[l], id[l] != 0));
        if (!fe_eq<P>(o.x, r.x) || !fe_eq<P>(o.y, r.y)) { printf("curve_tree: lane %d disagrees\n", l); abort(); }
    }
    for (int k = 0; k < 8; k++) { out[k] = r.x.l[k]; out[8 + k] = r.y.l[k]; }
}
extern "C" void hh_curve_tree(int curve, const uint32_t* bases, const uint32_t* signs, size_t n, int lanes, uint32_t* out) {
    if (curve == 0) curve_tree<PallasFp>(bases, signs, n, lanes, out);
    else curve_tree<PallasFq>(bases, signs, n, lanes, out);
}
// 1 when the bound assertions of field29.cuh / curve29.cuh are compiled in (they silently were not while another header pulled
// field29.cuh in ahead of the switch)
extern "C" int hh_f29_checks_active() { return F29_CHECKS_ACTIVE; }
extern "C" void hh_curve_sum(int curve, int mode, const uint32_t* bases, const uint32_t* signs, size_t n, uint32_t* out) {
    if (curve == 0) curve_sum<PallasFp>(mode, bases, signs, n, out);
    else curve_sum<PallasFq>(mode, bases, signs, n, out);
}


// ---------------------------------------------------------------------------------------------
#include "../../lurk_beta_amd/csrc/field29.cuh"
// radix-2^29 layer: op 0: to_mont256(from_mont256(a)); 1: mul; 2: add; 3: sub; 4: sub then sqr (loose operand,
// carried); 5: chain a*b - a + b via lazy ops.  Inputs/outputs are 8 x 32 Montgomery(2^256) values.
template <class P>
static void f29_ops(int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fe<P> x, y;
        for (int k = 0; k < 8; k++) { x.l[k] = a[8 * i + k]; y.l[k] = b[8 * i + k]; }
        F29<P> X = f29_from_mont256<P>(x), Y = f29_from_mont256<P>(y), Rz;
        switch (op) {
            case 0: Rz = X; break;
            case 1: Rz = f29_mul<P>(X, Y); break;
            case 2: Rz = f29_add<P>(X, Y); break;
            case 3: Rz = f29_sub<P>(X, Y); break;
            case 4: { F29<P> d = f29_carry<P>(f29_sub<P>(X, Y)); Rz = f29_mul<P>(d, d); break; }
            default: { F29<P> m = f29_mul<P>(X, Y); Rz = f29_add<P>(f29_carry<P>(f29_sub<P>(m, X)), Y); break; }
        }
        Fe<P> r = f29_to_mont256<P>(Rz);
        for (int k = 0; k < 8; k++) o[8 * i + k] = r.l[k];
    }
}
extern "C" void hh_f29_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
    if (field == 0) f29_ops<PallasFp>(op, a, b, o, n);
    else if (field == 1) f29_ops<PallasFq>(op, a, b, o, n);
    else f29_ops<Bn254Fr>(op, a, b, o, n);
}

// f29_reduce on raw limb patterns: in = 9 limbs (tight, top limb arbitrary); out = 9 limbs
extern "C" void hh_f29_reduce(int field, const uint32_t* in, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (field == 0) { F29<PallasFp> v; for (int k = 0; k < 9; k++) v.l[k] = in[9 * i + k]; v = f29_reduce<PallasFp>(v); for (int k = 0; k < 9; k++) out[9 * i + k] = v.l[k]; }
        else { F29<PallasFq> v; for (int k = 0; k < 9; k++) v.l[k] = in[9 * i + k]; v = f29_reduce<PallasFq>(v); for (int k = 0; k < 9; k++) out[9 * i + k] = v.l[k]; }
    }
}

#include "../../lurk_beta_amd/csrc/msm_core.cuh"
// signed c-bit digits of a canonical scalar two ways: the indexed recoding (msm_digit_step) and the register walk the sort kernels
// and the small path use (msm_digit_next); out: W digits each, |d| | sign << 31
extern "C" void hh_msm_digits(const uint32_t* scalar8, int c, uint32_t* by_step, uint32_t* by_walk) {
    const int W = msm_num_windows(c);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) by_step[w] = msm_digit_step(scalar8, w, c, carry);
    uint32_t r[8];
    for (int k = 0; k < 8; k++) r[k] = scalar8[k];
    carry = 0;
    for (int w = 0; w < W; w++) by_walk[w] = msm_digit_next(r, c, carry);
}

// the compile-time recoding of the sort kernels (msm_digits_ct<C>: all W digits at constant bit positions); returns W, or 0 for a width
// that has no instantiation here
template <int C>
static int hh_digits_ct(const uint32_t* scalar8, uint32_t* out) {
    uint32_t s[8], d[(256 + C - 1) / C];
    for (int k = 0; k < 8; k++) s[k] = scalar8[k];
    msm_digits_ct<C>(s, d);
    for (int w = 0; w < (256 + C - 1) / C; w++) out[w] = d[w];
    return (256 + C - 1) / C;
}
extern "C" int hh_msm_digits_ct(const uint32_t* scalar8, int c, uint32_t* out) {
    switch (c) {
        case 6: return hh_digits_ct<6>(scalar8, out);
        case 8: return hh_digits_ct<8>(scalar8, out);
        case 13: return hh_digits_ct<13>(scalar8, out);
        case 16: return hh_digits_ct<16>(scalar8, out);
        case 17: return hh_digits_ct<17>(scalar8, out);
        case 20: return hh_digits_ct<20>(scalar8, out);
        default: return 0;
    }
}

// ---------------------------------------------------------------------------------------------
// One sum-check round on host tables (sumcheck_host.hpp: what lurk_hip_sumcheck_prove_dev runs once the tables are short, and the
// arithmetic of the round kernel): tabs = np tables of len Montgomery elements, bound in place when r8 != NULL (the first len / 2
// elements of each are the new tables); ev = the evaluation sums at 0, 2 (, 3); returns the new length.
#include "../../lurk_beta_amd/csrc/sumcheck_host.hpp"
template <class F>
static size_t sumcheck_round_host(int np, uint32_t* tabs, size_t len, const uint32_t* r8, uint32_t* ev24) {
    std::vector<Fe<F>> P[4];
    for (int k = 0; k < np; k++) {
        P[k].resize(len);
        for (size_t i = 0; i < len; i++)
            for (int w = 0; w < 8; w++) P[k][i].l[w] = tabs[((size_t)k * len + i) * 8 + w];
    }
    Fe<F> r, ev[3];
    if (r8)
        for (int w = 0; w < 8; w++) r.l[w] = r8[w];
    size_t n = len;
    sumcheck_host_round<F>(np, P, n, r8 ? &r : nullptr, ev);
    for (int k = 0; k < np; k++)
        for (size_t i = 0; i < n; i++)
            for (int w = 0; w < 8; w++) tabs[((size_t)k * len + i) * 8 + w] = P[k][i].l[w];
    for (int k = 0; k < (np == 4 ? 3 : 2); k++)
        for (int w = 0; w < 8; w++) ev24[8 * k + w] = ev[k].l[w];
    return n;
}
extern "C" size_t hh_sumcheck_round(int field, int np, uint32_t* tabs, size_t len, const uint32_t* r8, uint32_t* ev24) {
    if (field == 0) return sumcheck_round_host<PallasFp>(np, tabs, len, r8, ev24);
    if (field == 1) return sumcheck_round_host<PallasFq>(np, tabs, len, r8, ev24);
    return sumcheck_round_host<Bn254Fr>(np, tabs, len, r8, ev24);
}

// two signed sums of bases normalised with ONE inversion (curve.cuh: xyzz_pair_to_affine, the opening argument's L and R) and one by
// one (xyzz_to_affine): out = 4 affine Montgomery records (pair: A, B; single: A, B)
template <class P>
static void pair_to_affine(const uint32_t* ba, const uint32_t* sa, size_t na, const uint32_t* bb, const uint32_t* sb, size_t nb, uint32_t* out) {
    auto sum = [](const uint32_t* bases, const uint32_t* signs, size_t n) {
        Xyzz<P> acc = xyzz_identity<P>();
        for (size_t i = 0; i < n; i++) {
            Affine<P> a;
            for (int k = 0; k < 8; k++) { a.x.l[k] = bases[i * 16 + k]; a.y.l[k] = bases[i * 16 + 8 + k]; }
            xyzz_madd<P>(acc, a, signs[i] != 0);
        }
        return acc;
    };
    const Xyzz<P> A = sum(ba, sa, na), B = sum(bb, sb, nb);
    Affine<P> r[4];
    xyzz_pair_to_affine<P>(A, B, r[0], r[1]);
    r[2] = xyzz_to_affine<P>(A);
    r[3] = xyzz_to_affine<P>(B);
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 8; k++) { out[16 * j + k] = r[j].x.l[k]; out[16 * j + 8 + k] = r[j].y.l[k]; }
}
extern "C" void hh_pair_to_affine(int curve, const uint32_t* ba, const uint32_t* sa, size_t na, const uint32_t* bb, const uint32_t* sb, size_t nb, uint32_t* out) {
    if (curve == 0) pair_to_affine<PallasFp>(ba, sa, na, bb, sb, nb, out);
    else pair_to_affine<PallasFq>(ba, sa, na, bb, sb, nb, out);
}

// ---------------------------------------------------------------------------------------------
// One point's row of a window table (msm_precompute.cuh: the body of msm_precompute_kernel, doubling chain + Montgomery's trick on the
// radix-2^29 layer, bound assertions on): out = W affine Montgomery records, 2^(c w) P for w < W.  Also f29_invert against fe_inv.
#include "../../lurk_beta_amd/csrc/msm_precompute.cuh"
template <class P>
static void precompute_row(const uint32_t* base16, int c, int W, uint32_t* out) {
    Affine<P> a;
    for (int k = 0; k < 8; k++) { a.x.l[k] = base16[k]; a.y.l[k] = base16[8 + k]; }
    std::vector<Affine<P>> table(W);
    std::vector<F29<P>> scratch((size_t)(W > 1 ? W - 1 : 1) * MSM_PRE_SLOTS);
    msm_precompute_point<P>(a, 0, 1, c, W, table.data(), scratch.data());
    for (int w = 0; w < W; w++)
        for (int k = 0; k < 8; k++) { out[16 * w + k] = table[w].x.l[k]; out[16 * w + 8 + k] = table[w].y.l[k]; }
}
extern "C" void hh_msm_precompute_row(int curve, const uint32_t* base16, int c, int W, uint32_t* out) {
    if (curve == 0) precompute_row<PallasFp>(base16, c, W, out);
    else precompute_row<PallasFq>(base16, c, W, out);
}
template <class P>
static void f29_invert_vs_fe(const uint32_t* a8, uint32_t* out29, uint32_t* out32) {
    Fe<P> x;
    for (int k = 0; k < 8; k++) x.l[k] = a8[k];
    const Fe<P> r29 = f29_to_mont256<P>(f29_invert<P>(f29_from_mont256<P>(x))), r32 = fe_inv<P>(x);
    for (int k = 0; k < 8; k++) { out29[k] = r29.l[k]; out32[k] = r32.l[k]; }
}
extern "C" void hh_f29_invert(int field, const uint32_t* a8, uint32_t* out29, uint32_t* out32) {
    if (field == 0) f29_invert_vs_fe<PallasFp>(a8, out29, out32);
    else f29_invert_vs_fe<PallasFq>(a8, out29, out32);
}

// ---------------------------------------------------------------------------------------------
// The key fold's host plan (keyfold_plan.hpp, used by ipa.hip: key_fold): digits[(b * Wt + j) * U + u] = the signed sub-digit of weight b in
// table window j, slot u, READ BACK FROM THE SORTED LISTS the kernel walks (ord / dstart / ord_base: magnitude by position, sign in bit 31).
// Returns 0, or a negative code when the structure is inconsistent (an entry listed twice, a magnitude range out of order, ...).
#include "../../lurk_beta_amd/csrc/keyfold_plan.hpp"
extern "C" int hh_keyfold_plan(const uint64_t* weights, size_t T, size_t m, int kb, int Wt, int* out_U, int* out_groups, int* out_maxmag, int* off, int* wd,
                               int32_t* digits) {
    const lurk::KeyFoldPlan pl = lurk::keyfold_plan(weights, T, m, kb, Wt);
    if (!pl.ok) return 1;
    *out_U = pl.U;
    *out_groups = pl.groups;
    *out_maxmag = pl.maxmag;
    for (int u = 0; u < pl.U; u++) {
        off[u] = pl.off[u];
        wd[u] = pl.wd[u];
    }
    for (size_t i = 0; i < T * (size_t)Wt * pl.U; i++) digits[i] = 0;
    std::vector<char> seen(T * (size_t)Wt * pl.U, 0);
    for (int u = 0; u < pl.U; u++)
        for (int g = 0; g < pl.groups; g++) {
            const size_t sg = (size_t)u * pl.groups + g;
            const uint32_t* ds = pl.dstart.data() + sg * (pl.maxmag + 1);
            const uint32_t* ord = pl.ord.data() + pl.ord_base[sg];
            if (ds[0] != 0) return -1;
            if (pl.ord_base[sg] + ds[pl.maxmag] != pl.ord_base[sg + 1]) return -2;
            for (int k = 0; k < pl.maxmag; k++) {
                if (ds[k + 1] < ds[k]) return -3;
                for (uint32_t t = ds[k]; t < ds[k + 1]; t++) {
                    const uint32_t e = ord[t], jb = e & 0x7fffffffu;
                    const size_t j = pl.jlo[g] + jb / T, b = jb % T;
                    if ((int)j >= pl.jlo[g + 1]) return -4;
                    const size_t at = (b * Wt + j) * pl.U + u;
                    if (seen[at]) return -5;
                    seen[at] = 1;
                    const int mag = pl.maxmag - k;
                    if (mag > (1 << (pl.wd[u] - 1))) return -6;
                    digits[at] = (e >> 31) ? -mag : mag;
                }
            }
        }
    return 0;
}


// host_poseidon.hpp - the Poseidon permutation on the HOST (4 x 64-bit Montgomery limbs), shared by the transcript (transcript.hip:
// Nova's random oracle, width 25) and the store hydration (store.hip: DAG levels too narrow to be worth a kernel's dependency
// chain, widths 4, 5, 7, 9).  Host code only; the schedule and the constants are those of the device kernels
// (poseidon_params.hpp: full rounds, pre-sparse matrix, sparse partial rounds, full rounds), so the host and the device produce
// the same digests - tests/test_gpu_poseidon.py compares them on every arity.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "poseidon_params.hpp"

namespace lurk {

// ---- host field arithmetic for the sponge: 4 x 64-bit Montgomery limbs, lazy inner products -----------------------------------------
// The permutation at width 25 is ~8 000 products, 5 000 of them in the dense layers of the 8 full rounds; a row of a dense layer is
// an inner product of 25 terms, accumulated here as a 576-bit integer and reduced ONCE (16 + 400 word products instead of 800).
typedef unsigned __int128 u128;
struct H4 {
    uint64_t v[4];
};
struct HostField {
    uint64_t m[4], inv;  // modulus, -m^-1 mod 2^64
};
template <class P>
static HostField host_field() {
    HostField F;
    for (int i = 0; i < 4; i++) F.m[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    uint64_t ninv = (uint64_t)0 - (uint64_t)P::INV;  // m^-1 mod 2^32 in the low half
    ninv *= 2 - F.m[0] * ninv;                        // Newton step: exact mod 2^64
    F.inv = (uint64_t)0 - ninv;
    return F;
}
template <class P>
static H4 h4_from(const Fe<P>& x) {
    H4 r;
    for (int i = 0; i < 4; i++) r.v[i] = (uint64_t)x.l[2 * i] | ((uint64_t)x.l[2 * i + 1] << 32);
    return r;
}
template <class P>
static Fe<P> h4_to(const H4& x) {
    Fe<P> r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)x.v[i];
        r.l[2 * i + 1] = (uint32_t)(x.v[i] >> 32);
    }
    return r;
}
static inline bool h4_geq(const uint64_t* a, const uint64_t* m) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > m[i]) return true;
        if (a[i] < m[i]) return false;
    }
    return true;
}
static inline void h4_sub_m(uint64_t* a, const uint64_t* m) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a[i] - m[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline H4 h4_add(const HostField& F, const H4& a, const H4& b) {
    H4 r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.v[i] + b.v[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || h4_geq(r.v, F.m)) h4_sub_m(r.v, F.m);  // both inputs < m < 2^255: no carry out of 256 bits, but keep the test
    return r;
}
// acc (9 limbs) += a * b
static inline void h4_mac(uint64_t* acc, const H4& a, const H4& b) {
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.v[j] * b.v[i] + acc[i + j];
            acc[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int k = i + 4; c && k < 9; k++) {
            c += acc[k];
            acc[k] = (uint64_t)c;
            c >>= 64;
        }
    }
}
// Montgomery reduction of a 9-limb T < 2^576 with T / 2^256 < 2^62 m: (T + q m) / 2^256 mod m, fully reduced
static inline H4 h4_redc9(const HostField& F, uint64_t* acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t q = acc[i] * F.inv;
        u128 c = ((u128)q * F.m[0] + acc[i]) >> 64;
        c += (u128)q * F.m[1] + acc[i + 1];
        acc[i + 1] = (uint64_t)c;
        c = (c >> 64) + (u128)q * F.m[2] + acc[i + 2];
        acc[i + 2] = (uint64_t)c;
        c = (c >> 64) + (u128)q * F.m[3] + acc[i + 3];
        acc[i + 3] = (uint64_t)c;
        c >>= 64;
#pragma unroll
        for (int k = i + 4; k < 9; k++) {  // straight-line carry: no data-dependent exit
            c += acc[k];
            acc[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    uint64_t* t = acc + 4;  // 5 limbs: < T / 2^256 + m
    // quotient estimate against 2^254 <= the Pasta moduli (BN254's r is just below 2^254: the estimate is then low, the loop below fixes it)
    uint64_t q = (t[4] << 2) | (t[3] >> 62);
    if (q) {
        u128 c = 0, br = 0;
        uint64_t qm[5];
        for (int j = 0; j < 4; j++) {
            c += (u128)q * F.m[j];
            qm[j] = (uint64_t)c;
            c >>= 64;
        }
        qm[4] = (uint64_t)c;
        bool less = false;  // t < q m ?
        for (int j = 4; j >= 0; j--) {
            if (t[j] != qm[j]) { less = t[j] < qm[j]; break; }
        }
        if (less) {  // one too many: subtract (q - 1) m instead
            q--;
            c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)q * F.m[j];
                qm[j] = (uint64_t)c;
                c >>= 64;
            }
            qm[4] = (uint64_t)c;
        }
        for (int j = 0; j < 5; j++) {
            const u128 d = (u128)t[j] - qm[j] - br;
            t[j] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    while (t[4] || h4_geq(t, F.m)) {
        u128 br = 0;
        for (int j = 0; j < 5; j++) {
            const u128 d = (u128)t[j] - (j < 4 ? F.m[j] : 0) - br;
            t[j] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    H4 r;
    for (int i = 0; i < 4; i++) r.v[i] = t[i];
    return r;
}
// sum_{i < n} a[i] * b[i * bstride] as a 9-limb integer, by columns (product scanning): the three-word column accumulator stays in
// registers across all n terms, so a row of a dense layer costs its 16 n word products and nothing else
static inline void h4_dot_columns(uint64_t* out9, const H4* a, const H4* b, int n, int bstride) {
    u128 lo = 0;
    uint64_t hi = 0;
    for (int k = 0; k < 7; k++) {
        const int i0 = k > 3 ? k - 3 : 0, i1 = k < 3 ? k : 3;
        for (int t = 0; t < n; t++) {
            const uint64_t* x = a[t].v;
            const uint64_t* y = b[(size_t)t * bstride].v;
            for (int i = i0; i <= i1; i++) {
                const u128 p = (u128)x[i] * y[k - i];
                lo += p;
                hi += lo < p;
            }
        }
        out9[k] = (uint64_t)lo;
        lo = (lo >> 64) | ((u128)hi << 64);
        hi = 0;
    }
    out9[7] = (uint64_t)lo;
    out9[8] = (uint64_t)(lo >> 64);
}
// one product: coarsely integrated operand scanning (CIOS), the four rounds unrolled - a multiplication row and a reduction row per
// word of b, five live accumulator words, one conditional subtraction at the end (inputs < m < 2^255: the result before it is < 2 m).
// The S-boxes and the sparse partial rounds are chains of single products; through h4_mac + h4_redc9 (a 9-limb accumulator with
// carry loops and a quotient estimate, built for the 25-term rows) each cost ~33 ns, this form ~14.
static inline H4 h4_mul(const HostField& F, const H4& a, const H4& b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = b.v[i];
        u128 c = (u128)a.v[0] * bi + t0;
        t0 = (uint64_t)c;
        c = (c >> 64) + (u128)a.v[1] * bi + t1;
        t1 = (uint64_t)c;
        c = (c >> 64) + (u128)a.v[2] * bi + t2;
        t2 = (uint64_t)c;
        c = (c >> 64) + (u128)a.v[3] * bi + t3;
        t3 = (uint64_t)c;
        c = (c >> 64) + t4;
        t4 = (uint64_t)c;
        const uint64_t t5 = (uint64_t)(c >> 64);
        const uint64_t q = t0 * F.inv;
        c = ((u128)q * F.m[0] + t0) >> 64;
        c += (u128)q * F.m[1] + t1;
        t0 = (uint64_t)c;
        c = (c >> 64) + (u128)q * F.m[2] + t2;
        t1 = (uint64_t)c;
        c = (c >> 64) + (u128)q * F.m[3] + t3;
        t2 = (uint64_t)c;
        c = (c >> 64) + t4;
        t3 = (uint64_t)c;
        t4 = t5 + (uint64_t)(c >> 64);
    }
    H4 r = {{t0, t1, t2, t3}};
    if (t4 || h4_geq(r.v, F.m)) h4_sub_m(r.v, F.m);
    return r;
}
static inline H4 h4_pow5(const HostField& F, const H4& x) {
    const H4 x2 = h4_mul(F, x, x), x4 = h4_mul(F, x2, x2);
    return h4_mul(F, x4, x);
}

// the sponge's constants in host form (the sparse schedule of poseidon_params.hpp), built once per field
struct RoHost {
    HostField F;
    int t, rf, rp;
    std::vector<H4> rc, mds, pre_sparse, sparse, partial_k, rc_after;
    H4 r2;  // 2^512 mod m: canonical -> Montgomery
};
// one table per (field, arity), built on first use and kept for the life of the process
template <class P>
static const RoHost& poseidon_host(int arity) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<RoHost>> tables;
    std::lock_guard<std::mutex> lk(mu);
    std::unique_ptr<RoHost>& h = tables[arity];
    if (!h) {
        const PoseidonParams<P> pp = make_poseidon_params<P>(arity);
        h.reset(new RoHost());
        h->F = host_field<P>();
        h->t = pp.t;
        h->rf = pp.rf;
        h->rp = pp.rp;
        auto conv = [](const std::vector<Fe<P>>& v) {
            std::vector<H4> o;
            for (auto& x : v) o.push_back(h4_from<P>(x));
            return o;
        };
        h->rc = conv(pp.rc);
        h->mds = conv(pp.mds);
        h->pre_sparse = conv(pp.pre_sparse);
        h->sparse = conv(pp.sparse);
        h->partial_k = conv(pp.partial_k);
        h->rc_after = conv(pp.rc_after);
        h->r2 = h4_from<P>(fe_r2<P>());
    }
    return *h;
}

// u = Mat s, row-major (the Cauchy matrix is symmetric: the same as neptune's row-vector convention); one reduction per row
static void dense_host(const RoHost& R, std::vector<H4>& s, const H4* mat) {
    const int t = R.t;
    H4 u[32];  // widths in use: 4, 5, 7, 9, 25
    for (int j = 0; j < t; j++) {
        uint64_t acc[9];
        h4_dot_columns(acc, s.data(), mat + (size_t)j * t, t, 1);
        u[j] = h4_redc9(R.F, acc);
    }
    for (int j = 0; j < t; j++) s[j] = u[j];
}
// the same schedule as poseidon.cuh: poseidon_permute (full rounds, pre-sparse matrix, sparse partial rounds, full rounds)
static void permute_host(const RoHost& R, std::vector<H4>& s) {
    const int t = R.t, h = R.rf / 2;
    for (int r = 0; r < h; r++) {
        for (int i = 0; i < t; i++) s[i] = h4_pow5(R.F, h4_add(R.F, s[i], R.rc[(size_t)r * t + i]));
        dense_host(R, s, r == h - 1 ? R.pre_sparse.data() : R.mds.data());
    }
    for (int p = 0; p < R.rp; p++) {
        const H4* sp = &R.sparse[(size_t)p * (2 * t - 1)];
        const H4 x = h4_pow5(R.F, h4_add(R.F, s[0], R.partial_k[p]));
        const H4 s0 = s[0];
        s[0] = x;  // the row (x n00 | s_i v_i) as one inner product over the state with x in front
        uint64_t acc[9];
        h4_dot_columns(acc, s.data(), sp, t, 1);
        for (int i = 1; i < t; i++) s[i] = h4_add(R.F, s[i], h4_mul(R.F, x, sp[t - 1 + i]));
        (void)s0;
        s[0] = h4_redc9(R.F, acc);
    }
    for (int r = 0; r < h; r++) {
        const H4* rc = r == 0 ? R.rc_after.data() : &R.rc[(size_t)(h + R.rp + r) * t];
        for (int i = 0; i < t; i++) s[i] = h4_pow5(R.F, h4_add(R.F, s[i], rc[i]));
        dense_host(R, s, R.mds.data());
    }
}

// neptune's fixed-arity hash (PoseidonCache::hash3/4/6/8, /root/reference/src/hash.rs:180-204): state = [2^arity - 1 | preimage], one
// permutation, digest = state[1].  Canonical 32-byte elements in and out.
template <class P>
static void poseidon_hash_host(const RoHost& R, const uint64_t* pre, uint64_t* out4) {
    const int t = R.t, arity = t - 1;
    const H4 one = {{1, 0, 0, 0}};
    std::vector<H4> s(t);
    const H4 tag = {{((uint64_t)1 << arity) - 1, 0, 0, 0}};
    s[0] = h4_mul(R.F, tag, R.r2);
    for (int i = 0; i < arity; i++) {
        H4 e;
        memcpy(e.v, pre + 4 * i, 32);
        LURK_REQUIRE(!h4_geq(e.v, R.F.m), "Poseidon preimage element is not a canonical field element");
        s[1 + i] = h4_mul(R.F, e, R.r2);
    }
    permute_host(R, s);
    const H4 d = h4_mul(R.F, s[1], one);
    memcpy(out4, d.v, 32);
}

}  // namespace lurk

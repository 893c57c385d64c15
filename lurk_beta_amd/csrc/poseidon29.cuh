// poseidon29.cuh - the Poseidon permutation on the radix-2^29 layer (field29.cuh): what the gfx950
// kernels run.  Same schedule, same constants and same results as poseidon.cuh (which stays as the
// 8 x 32-bit statement of the algorithm that tests/host_harness checks this one against); only the
// arithmetic representation differs: nine 29-bit limbs, Montgomery radix 2^261, lazy reduction, no carry
// folds (gfx950 issues a carry fold almost as slowly as the v_mad_u64_u32 it follows).
//
// Constant image: poseidon_device_image's elements in the same order, each as 12 words: nine limbs of
// the CANONICAL value c * 2^261 mod p (< p < 2^254.1) + 3 words of padding (16-byte aligned LDS reads);
// one extra element at the end: 2^522 mod p (turns a plain value into its Montgomery form).
//
// Bounds (asserted in the host-harness build):
//   constants  tight, < 2^254.1
//   state      tight limbs, value < 2^260.1   (rows: < T * 2^253 + p; partial rounds add < 2^254.1 each)
//   rows       sum_i s_i * c_i / 2^261: columns of up to T * 9 products of < 2^58: the 17 column
//              accumulators are normalised once after the 4th operand when T > 5 (45 products fit 64 bits)
#pragma once
#include <vector>

#include "field29.cuh"
#include "poseidon.cuh"

namespace lurk {

constexpr int P29_STRIDE = 12;  // words per constant in the image

template <class P>
LURK_HD F29<P> ld_const29(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(p));  // opaque per use: keeps the (round-invariant) matrix loads inside the round loops
#endif
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
}

template <class P>
LURK_HD F29<P> f29_pow5(const F29<P>& x) {  // x: limbs < 2^30
    F29<P> x2 = f29_sqr<P>(x);
    F29<P> x4 = f29_sqr<P>(x2);
    return f29_mul<P>(x4, x);
}

// s <- Mat * s (dense row-major T x T)
template <class P, int T>
LURK_HD void poseidon29_dense(F29<P>* s, const uint32_t* mat) {
    F29<P> u[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        Dot29<P> A;
        dot29_init<P>(A);
#pragma unroll
        for (int i = 0; i < T; i++) {
            if (T > 5 && i == 4) dot29_carry<P>(A);
            dot29_mac<P>(A, s[i], ld_const29<P>(mat + (j * T + i) * P29_STRIDE));
        }
        u[j] = dot29_finish<P>(A);
    }
#pragma unroll
    for (int j = 0; j < T; j++) s[j] = u[j];
}

template <class P, int T>
LURK_HD void poseidon29_full_round(F29<P>* s, const uint32_t* rc, const uint32_t* mat) {
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = f29_pow5<P>(f29_add<P>(s[i], ld_const29<P>(rc + i * P29_STRIDE)));
    poseidon29_dense<P, T>(s, mat);
}

// Sparse-schedule permutation over the radix-2^29 image (layout: PoseidonLayout, element e at img + e * 12).
template <class P, int T>
LURK_HD void poseidon29_permute(F29<P>* s, const uint32_t* img, int rf, int rp) {
    const PoseidonLayout<T> L(rf, rp);
    const uint32_t* mds = img + L.mds() * P29_STRIDE;
#pragma unroll 1
    for (int r = 0; r < L.h; r++)
        poseidon29_full_round<P, T>(s, img + (L.rc1() + r * T) * P29_STRIDE, r == L.h - 1 ? img + L.pre() * P29_STRIDE : mds);
#pragma unroll 1
    for (int p = 0; p < rp; p++) {
        const uint32_t* sp = img + (L.sp() + p * (2 * T - 1)) * P29_STRIDE;
        F29<P> x = f29_pow5<P>(f29_add<P>(s[0], ld_const29<P>(img + (L.pk() + p) * P29_STRIDE)));
        Dot29<P> A;
        dot29_init<P>(A);
        dot29_mac<P>(A, x, ld_const29<P>(sp));
#pragma unroll
        for (int i = 1; i < T; i++) {
            if (T > 5 && i == 4) dot29_carry<P>(A);
            dot29_mac<P>(A, s[i], ld_const29<P>(sp + i * P29_STRIDE));
            // s_i += x * w_i, limbs brought back under 2^29 (the value grows by < 2^254.1 per round)
            s[i] = f29_carry<P>(f29_add<P>(s[i], f29_mul<P>(x, ld_const29<P>(sp + (T - 1 + i) * P29_STRIDE))));
        }
        s[0] = dot29_finish<P>(A);
    }
#pragma unroll 1
    for (int r = 0; r < L.h; r++)
        poseidon29_full_round<P, T>(s, r == 0 ? img + L.after() * P29_STRIDE : img + (L.rc2() + (r - 1) * T) * P29_STRIDE, mds);
}

// The same permutation with the circuit's S-box witnesses handed to `emit(sbox, l2, l4, l5k)` (lazy radix-2^29 values):
// sbox = position of the S-box in circuit order, l2 = l^2, l4 = l^4, l5k = l^5 + post key (neptune circuit2 allocates
// exactly these three per S-box).  post: the post keys, one image record per S-box.
template <class P, class E>
LURK_HD F29<P> f29_pow5_trace(const F29<P>& l, const uint32_t* post, int sbox, E& emit) {
    const F29<P> l2 = f29_sqr<P>(l);
    const F29<P> l4 = f29_sqr<P>(l2);
    const F29<P> l5 = f29_mul<P>(l4, l);
    emit(sbox, l2, l4, f29_add<P>(l5, ld_const29<P>(post + (size_t)sbox * P29_STRIDE)));
    return l5;
}
template <class P, int T, class E>
LURK_HD void poseidon29_permute_trace(F29<P>* s, const uint32_t* img, const uint32_t* post, int rf, int rp, E& emit) {
    const PoseidonLayout<T> L(rf, rp);
    const uint32_t* mds = img + L.mds() * P29_STRIDE;
    int sbox = 0;
    auto full_round = [&](const uint32_t* rc, const uint32_t* mat) {
        for (int i = 0; i < T; i++) s[i] = f29_pow5_trace<P>(f29_carry<P>(f29_add<P>(s[i], ld_const29<P>(rc + i * P29_STRIDE))), post, sbox++, emit);
        poseidon29_dense<P, T>(s, mat);
    };
    for (int r = 0; r < L.h; r++) full_round(img + (L.rc1() + r * T) * P29_STRIDE, r == L.h - 1 ? img + L.pre() * P29_STRIDE : mds);
    for (int p = 0; p < rp; p++) {
        const uint32_t* sp = img + (L.sp() + p * (2 * T - 1)) * P29_STRIDE;
        F29<P> x = f29_pow5_trace<P>(f29_carry<P>(f29_add<P>(s[0], ld_const29<P>(img + (L.pk() + p) * P29_STRIDE))), post, sbox++, emit);
        Dot29<P> A;
        dot29_init<P>(A);
        dot29_mac<P>(A, x, ld_const29<P>(sp));
        for (int i = 1; i < T; i++) {
            if (T > 5 && i == 4) dot29_carry<P>(A);
            dot29_mac<P>(A, s[i], ld_const29<P>(sp + i * P29_STRIDE));
            s[i] = f29_carry<P>(f29_add<P>(s[i], f29_mul<P>(x, ld_const29<P>(sp + (T - 1 + i) * P29_STRIDE))));
        }
        s[0] = dot29_finish<P>(A);
    }
    for (int r = 0; r < L.h; r++) full_round(r == 0 ? img + L.after() * P29_STRIDE : img + (L.rc2() + (r - 1) * T) * P29_STRIDE, mds);
}

// plain canonical value (8 x 32) -> state element; mont2 = the image's last element (2^522 mod p)
template <class P>
LURK_HD F29<P> poseidon29_from_canonical(const uint32_t* x, const uint32_t* mont2) {
    return f29_mul<P>(f29_from_plain<P>(x), ld_const29<P>(mont2));
}
// state element -> plain canonical value
template <class P>
LURK_HD Fe<P> poseidon29_to_canonical(const F29<P>& a) {
    F29<P> one = f29_zero<P>();
    one.l[0] = 1;
    F29<P> u = f29_mul<P>(f29_carry<P>(a), one);  // a / 2^261 mod p, < p + 1
    uint32_t w[8];
    f29_pack<P>(u, w);
    fe_cond_sub<P>(w);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
    return r;
}

// host: re-express the 8 x 32 image (canonical Montgomery-2^256 elements) in the radix-2^29 form
// extra256: further canonical Montgomery-2^256 elements appended AFTER the 2^522 record (the trace kernels' post keys)
template <class P>
std::vector<uint32_t> poseidon29_image(const std::vector<uint32_t>& img256, const std::vector<uint32_t>& extra256 = {}) {
    const size_t n = img256.size() / 8, m = extra256.size() / 8;
    std::vector<uint32_t> out((n + 1 + m) * P29_STRIDE, 0u);
    auto put = [&](size_t e, const Fe<P>& v) {  // v: canonical value to be stored as plain limbs
        F29<P> f = f29_from_plain<P>(v.l);
        for (int i = 0; i < 9; i++) out[e * P29_STRIDE + i] = f.l[i];
    };
    for (size_t e = 0; e < n; e++) {
        Fe<P> v;
        for (int i = 0; i < 8; i++) v.l[i] = img256[e * 8 + i];
        for (int d = 0; d < 5; d++) v = fe_add<P>(v, v);  // c * 2^256 -> c * 2^261 (mod p, canonical)
        put(e, v);
    }
    // 2^522 mod p = mont256(2^266) -> as a plain integer: (R2 * R2 / R) is 2^512 / 2^256 ... build it by doublings
    Fe<P> v = fe_one<P>();                       // 2^256 mod p as a plain integer
    for (int d = 0; d < 522 - 256; d++) v = fe_add<P>(v, v);
    put(n, v);
    for (size_t e = 0; e < m; e++) {
        Fe<P> x;
        for (int i = 0; i < 8; i++) x.l[i] = extra256[e * 8 + i];
        for (int d = 0; d < 5; d++) x = fe_add<P>(x, x);
        put(n + 1 + e, x);
    }
    return out;
}

}  // namespace lurk

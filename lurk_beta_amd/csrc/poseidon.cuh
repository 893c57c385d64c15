// poseidon.cuh - the Poseidon permutation as run by the gfx950 kernels: one state per lane, the
// constant image (poseidon_params.hpp: poseidon_device_image) read through a wave-uniform pointer
// (LDS in the kernels).  Shared host/device so tests/host_harness can check the exact code path
// against the oracle on the CPU.
//
// Reference semantics: Poseidon::new_with_preimage(preimage, consts).hash()
// (/root/reference/src/hash.rs:181-203): state = [domain_tag, preimage...], R_F/2 full rounds,
// R_P partial rounds, R_F/2 full rounds, digest = state[1].
#pragma once
#include "field.cuh"

namespace lurk {

template <int T>
struct PoseidonLayout {  // offsets in field elements into the device image
    int h, rp;
    LURK_HD PoseidonLayout(int rf, int rp_) : h(rf / 2), rp(rp_) {}
    LURK_HD int tag() const { return 0; }
    LURK_HD int rc1() const { return 1; }
    LURK_HD int mds() const { return 1 + h * T; }
    LURK_HD int pre() const { return mds() + T * T; }
    LURK_HD int pk() const { return pre() + T * T; }
    LURK_HD int sp() const { return pk() + rp; }
    LURK_HD int after() const { return sp() + rp * (2 * T - 1); }
    LURK_HD int rc2() const { return after() + T; }
    LURK_HD int total() const { return rc2() + (h - 1) * T; }
};

// Constant fetch.  On the device the address is made opaque per use so that the compiler neither
// hoists the (round-invariant) MDS loads out of the round loops nor batches a whole matrix of
// ds_reads ahead of the multiplier calls: either would need hundreds of VGPRs and spill.
template <class P>
LURK_HD Fe<P> ld_const(const Fe<P>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(p));
#endif
    return *p;
}

template <class P>
LURK_HD Fe<P> fe_pow5(const Fe<P>& x) {
    Fe<P> x2 = fe_sqr<P>(x);
    Fe<P> x4 = fe_sqr<P>(x2);
    return fe_mul<P>(x4, x);
}

// s <- Mat * s for a dense row-major T x T matrix (Cauchy MDS is symmetric, so this equals
// neptune's row-vector convention)
template <class P, int T>
LURK_HD void poseidon_dense(Fe<P>* s, const Fe<P>* mat) {
    Fe<P> u[T];
#pragma unroll
    for (int j = 0; j < T; j++) {
        // one row = one inner product with a single Montgomery reduction (field.cuh: dot_mac)
        DotAcc<P> A;
        dot_init<P>(A);
#pragma unroll
        for (int i = 0; i < T; i++) dot_mac<P>(A, s[i], ld_const<P>(mat + j * T + i));
        u[j] = dot_finish<P, T>(A);
    }
#pragma unroll
    for (int j = 0; j < T; j++) s[j] = u[j];
}

template <class P, int T>
LURK_HD void poseidon_full_round(Fe<P>* s, const Fe<P>* rc, const Fe<P>* mat) {
#pragma unroll
    for (int i = 0; i < T; i++) s[i] = fe_pow5<P>(fe_add<P>(s[i], ld_const<P>(rc + i)));
    poseidon_dense<P, T>(s, mat);
}

// Sparse-schedule permutation over the device image `img` (see poseidon_params.hpp).
template <class P, int T>
LURK_HD void poseidon_permute(Fe<P>* s, const Fe<P>* img, int rf, int rp) {
    const PoseidonLayout<T> L(rf, rp);
    const Fe<P>* mds = img + L.mds();
    // full rounds 0 .. h-1 (the last one multiplies by the pre-sparse matrix instead of the MDS)
#pragma unroll 1
    for (int r = 0; r < L.h; r++)
        poseidon_full_round<P, T>(s, img + L.rc1() + r * T, r == L.h - 1 ? img + L.pre() : mds);
#pragma unroll 1
    for (int p = 0; p < rp; p++) {
        const Fe<P>* sp = img + L.sp() + p * (2 * T - 1);
        Fe<P> x = fe_pow5<P>(fe_add<P>(s[0], ld_const<P>(img + L.pk() + p)));
        DotAcc<P> A;
        dot_init<P>(A);
        dot_mac<P>(A, x, ld_const<P>(sp));
#pragma unroll
        for (int i = 1; i < T; i++) {
            dot_mac<P>(A, s[i], ld_const<P>(sp + i));
            s[i] = fe_add<P>(s[i], fe_mul<P>(x, ld_const<P>(sp + T - 1 + i)));
        }
        s[0] = dot_finish<P, T>(A);
    }
    // full rounds h+rp .. rf+rp-1 (the first one takes the constants that absorbed the partial rounds')
#pragma unroll 1
    for (int r = 0; r < L.h; r++)
        poseidon_full_round<P, T>(s, r == 0 ? img + L.after() : img + L.rc2() + (r - 1) * T, mds);
}

// Plain schedule over (rc, mds) - used by tests to cross-check the sparse derivation.
template <class P, int T>
LURK_HD void poseidon_permute_plain(Fe<P>* s, const Fe<P>* rc, const Fe<P>* mds, int rf, int rp) {
    for (int r = 0; r < rf + rp; r++) {
        bool full = r < rf / 2 || r >= rf / 2 + rp;
        if (full) {
            poseidon_full_round<P, T>(s, rc + r * T, mds);
        } else {
            s[0] = fe_pow5<P>(fe_add<P>(s[0], rc[r * T]));
            for (int i = 1; i < T; i++) s[i] = fe_add<P>(s[i], rc[r * T + i]);
            poseidon_dense<P, T>(s, mds);
        }
    }
}

}  // namespace lurk

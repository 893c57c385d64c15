// msm_precompute.cuh - one base's row of the window table, T[w n + i] = 2^(c w) P_i: the body of msm_precompute_kernel
// (msm_precompute.hip), host-callable so that the harness runs it with the radix-2^29 bound assertions on.
//
// One inversion per POINT, not per table entry: the W - 1 multiples are carried in XYZZ form, their (X, Y) parked in the table,
// ZZ, ZZZ and the running product of the ZZZ parked in a scratch buffer, then one field inversion and Montgomery's trick walk
// back over the windows (x = X / ZZ, y = Y / ZZZ, 1 / ZZ = (ZZ / ZZZ)^2).
//
// Round 5: the whole chain runs on the radix-2^29 layer.  A lane walks c (W - 1) DEPENDENT doublings, so what a key's table costs
// is the latency of one field product on a lane; the 8 x 32 Montgomery product (mad + carry-fold pairs, called as a function)
// takes ~1.7 us there, the carry-free 9 x 29 one ~0.45.  The opening argument folds its key once per proof and needs the folded
// key's table (65 536 points, 240 doublings each): 4.35 ms of every 30 ms proof before, see DESIGN.md section 3.7.
#pragma once
#include "curve29.cuh"

namespace lurk {

// F29 records of the scratch buffer per (window, point): ZZ, ZZZ, product of ZZZ_1..w
constexpr int MSM_PRE_SLOTS = 3;
size_t msm_precompute_scratch_bytes(size_t n, int W);  // (W - 1) x MSM_PRE_SLOTS x n records of 36 B

// table: W rows of `n` records, this point's column i; scratch: (W - 1) x 3 rows of n F29 records, this point's column i
template <class P>
LURK_HD void msm_precompute_point(const Affine<P>& a, size_t i, size_t n, int c, int W, Affine<P>* table, F29<P>* scratch) {
    table[i] = a;
    if (affine_is_identity<P>(a)) {
        for (int w = 1; w < W; w++) table[(size_t)w * n + i] = a;
        return;
    }
    auto slot = [&](int w, int k) -> F29<P>& { return scratch[((size_t)(w - 1) * MSM_PRE_SLOTS + k) * n + i]; };
    const F29<P> one = f29_from_mont256<P>(fe_one<P>());
    Xyzz29<P> p;
    p.x = f29_from_mont256<P>(a.x);
    p.y = f29_from_mont256<P>(a.y);
    p.zz = one;
    p.zzz = one;
    F29<P> prod = one;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int w = 1; w < W; w++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int k = 0; k < c; k++) p = xyzz29_dbl<P>(p);
        table[(size_t)w * n + i] = Affine<P>{f29_to_mont256<P>(p.x), f29_to_mont256<P>(p.y)};  // parked: X, Y (canonical Montgomery 2^256)
        prod = f29_mul<P>(prod, p.zzz);
        slot(w, 0) = p.zz;
        slot(w, 1) = p.zzz;
        slot(w, 2) = prod;
    }
    F29<P> inv = f29_invert<P>(prod);  // 1 / (ZZZ_1 ... ZZZ_{W-1})
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int w = W - 1; w >= 1; w--) {
        const F29<P> zzz = slot(w, 1);
        const F29<P> zzz_inv = w > 1 ? f29_mul<P>(inv, slot(w - 1, 2)) : inv;
        inv = f29_mul<P>(inv, zzz);
        const F29<P> t = f29_mul<P>(slot(w, 0), zzz_inv);  // ZZ / ZZZ = 1 / Z
        const F29<P> zz_inv = f29_sqr<P>(t);
        const Affine<P> q = table[(size_t)w * n + i];
        const F29<P> x = f29_mul<P>(f29_from_mont256<P>(q.x), zz_inv);
        const F29<P> y = f29_mul<P>(f29_from_mont256<P>(q.y), zzz_inv);
        table[(size_t)w * n + i] = Affine<P>{f29_to_mont256<P>(x), f29_to_mont256<P>(y)};
    }
}

#if defined(__HIPCC__)
// enqueues the table build on s (scratch: msm_precompute_scratch_bytes(n, W), free again when the launch has completed)
template <class P>
void msm_launch_precompute(const Affine<P>* bases, size_t n, Affine<P>* table, int c, int W, void* scratch, hipStream_t s);
#endif

}  // namespace lurk

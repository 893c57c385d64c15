// msm_core.cuh - per-thread building blocks of the Pippenger MSM (shared host/device; the kernels
// in msm.hip call these with their thread/block indices, tests/host_harness replays the same
// pipeline serially on the CPU).
//
// Pipeline (c = 16-bit signed windows, W = 16 windows, B = 2^15 buckets per key space):
//   1. digits      scalar -> 16 signed digits d_w in [-2^15, 2^15]           (msm_scalar_digits)
//   2. sort        counting sort of (point, sign) entries by bucket, per key space; LDS-privatised
//                  histograms, no global atomics                                (msm.hip)
//   3. accumulate  buckets are cut into tasks of <= S sorted entries; one lane per task runs
//                  XYZZ mixed additions over gathered bases                    (msm_task_accumulate)
//   4. finalize    per bucket: sum of its task partials (hot buckets by a workgroup tree)
//   5. reduce      sum_b b*B_b = S + sum_k 2^k P_k: bit-plane merge tree (15 levels of depth one
//                  addition) + tree-shaped Horner                              (msm.hip)
//   6. combine     sum_w 2^(16w) * W_w : 240 sequential doublings -> done on the host over 16
//                  points (a latency-bound tail; with the precomputed table G = 1 and it vanishes)
// "Key space" = set of buckets entries are sorted into: one per window in the generic mode
// (G = 16); a single shared one (G = 1) when the context holds the precomputed table
// T[w*n + i] = 2^(16w) * P_i, because then every window's entry refers to its own table row.
#pragma once
#include "curve.cuh"

namespace lurk {

constexpr int MSM_C = 16;                      // window bits
constexpr int MSM_W = 16;                      // windows (16*16 = 256 >= 255 + carry)
constexpr int MSM_B = 1 << (MSM_C - 1);        // buckets per key space (|digit| in 1..2^15)
constexpr uint32_t MSM_SIGN = 0x80000000u;

// Signed-digit recoding of a canonical 255-bit scalar (8 x u32 LE): out[w] = |d_w| | sign<<31.
LURK_HD void msm_scalar_digits(const uint32_t* s, uint32_t* out) {
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < MSM_W; w++) {
        uint32_t raw = ((s[w >> 1] >> ((w & 1) * 16)) & 0xffffu) + carry;
        if (raw > (uint32_t)MSM_B) {
            out[w] = (0x10000u - raw) | MSM_SIGN;
            carry = 1;
        } else {
            out[w] = raw;
            carry = 0;
        }
    }
    // scalars are < 2^255: the top window never carries out
}

// largest g with start[g] <= t, over start[0..n] (start[n] is the sentinel = total)
LURK_HD uint32_t msm_upper_slot(const uint32_t* start, uint32_t n, uint32_t t) {
    uint32_t lo = 0, hi = n;  // invariant: start[lo] <= t < start[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

// One accumulation task: entries [first, last) of the sorted list, all of one bucket.
template <class P>
LURK_HD Xyzz<P> msm_task_accumulate(const uint32_t* sorted, uint32_t first, uint32_t last, const Affine<P>* table) {
    Xyzz<P> acc = xyzz_identity<P>();
    for (uint32_t j = first; j < last; j++) {
        uint32_t e = sorted[j];
        Affine<P> q = table[e & ~MSM_SIGN];
        xyzz_madd<P>(acc, q, (e & MSM_SIGN) != 0);
    }
    return acc;
}

template <class P>
LURK_HD Xyzz<P> xyzz_dbl_n(Xyzz<P> p, int n) {
    for (int i = 0; i < n; i++) p = xyzz_dbl<P>(p);
    return p;
}

// Host tail: sum_g 2^(16 g) * ws[g], Horner from the top window.
template <class P>
LURK_HD Xyzz<P> msm_combine_windows(const Xyzz<P>* ws, int g) {
    Xyzz<P> acc = xyzz_identity<P>();
    for (int w = g - 1; w >= 0; w--) {
        acc = xyzz_dbl_n<P>(acc, MSM_C);
        xyzz_add<P>(acc, ws[w]);
    }
    return acc;
}

}  // namespace lurk

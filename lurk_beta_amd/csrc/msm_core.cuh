// msm_core.cuh - per-thread building blocks of the Pippenger MSM (shared host/device; the kernels
// in msm.hip call these with their thread/block indices, tests/host_harness replays the same
// pipeline serially on the CPU).
//
// Pipeline (c-bit signed windows, W = ceil(256/c) windows, B = 2^(c-1) buckets per key space):
//   1. digits      scalar -> W signed digits d_w in [-2^(c-1), 2^(c-1)]        (msm_scalar_digits)
//   2. sort        two-pass partitioned counting sort of (point, sign) entries by key
//                  (key = space*B + |d|-1): 2048 coarse partitions, then the low key bits inside
//                  each partition; LDS counters only, both passes stage their output in LDS and
//                  write it back in runs                                        (msm.hip)
//   3. accumulate  buckets are cut into tasks of <= S sorted entries, ordered longest first; one
//                  lane per task runs XYZZ mixed additions over gathered bases on the radix-2^29
//                  layer (curve29.cuh: msm_task_accumulate29; msm_task_accumulate below is the
//                  32-bit-limb statement of the same loop, kept for A/B runs and the host harness)
//   4. finalize    per bucket: sum of its task partials (hot buckets by a workgroup tree)
//   5. reduce      sum_b b*B_b = S + sum_k 2^k P_k: bit-plane merge tree (c-1 levels of depth one
//                  addition) + Horner                                           (msm.hip)
//   6. combine     sum_w 2^(c*w) * W_w on the host over <= 16 points (sequential doublings: a
//                  latency-bound tail that one host core runs ~20x faster than one GPU lane)
// "Key space" = set of buckets entries are sorted into: one per window in the plain mode (G = W,
// c = 16); a single shared one (G = 1, c = 18 or 20) when the context holds the precomputed table
// T[w*n + i] = 2^(c*w) * P_i, because then every window's entry refers to its own table row and
// the wider window costs no extra bucket sets (13 n mixed additions instead of 16 n at c = 20).
#pragma once
#include "curve.cuh"

namespace lurk {

constexpr int MSM_C_PLAIN = 16;                // window bits of the plain mode
constexpr int MSM_MAX_W = 16;                  // windows: ceil(256 / c), c >= 16
constexpr int MSM_GRP = 1 << 15;               // keys per scan group
constexpr uint32_t MSM_SIGN = 0x80000000u;
constexpr int MSM_PLACEMENT_BASE = 16;         // word offset of the per-CU workgroup counters behind the persistent kernel's task cursor

LURK_HD int msm_num_windows(int c) { return (256 + c - 1) / c; }

// Signed-digit recoding of a canonical 255-bit scalar (8 x u32 LE) with c-bit windows, one window
// per call (w ascending, carry threaded through): returns |d_w| | sign<<31 with |d_w| <= 2^(c-1).
// Scalars are < 2^255 and W*c >= 256, so the top window never carries out.
LURK_HD uint32_t msm_digit_step(const uint32_t* s, int w, int c, uint32_t& carry) {
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1u;
    const int bit = w * c, limb = bit >> 5, sh = bit & 31;
    uint32_t raw = 0;
    if (limb < 8) {
        uint64_t v = s[limb];
        if (limb + 1 < 8) v |= (uint64_t)s[limb + 1] << 32;
        raw = (uint32_t)(v >> sh) & mask;
    }
    raw += carry;
    if (raw > half) {
        carry = 1;
        return ((1u << c) - raw) | MSM_SIGN;
    }
    carry = 0;
    return raw;
}

// The same recoding as a WALK over the windows in ascending order: the scalar sits in r[0..8) and is consumed from the bottom (the
// registers are shifted down by c bits per digit), so no limb is ever indexed by a run-time value - with msm_digit_step's s[limb]
// the compiler keeps the scalar in scratch memory and every digit costs two scratch loads.  1 <= c <= 31; past the top window the
// registers are empty and the carry is 0: the digits are 0.
LURK_HD uint32_t msm_digit_next(uint32_t (&r)[8], int c, uint32_t& carry) {
    const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1u;
    uint32_t raw = (r[0] & mask) + carry;
#pragma unroll
    for (int k = 0; k < 7; k++) r[k] = (r[k] >> c) | (r[k + 1] << (32 - c));
    r[7] >>= c;
    if (raw > half) {
        carry = 1;
        return ((1u << c) - raw) | MSM_SIGN;
    }
    carry = 0;
    return raw;
}

// All W = ceil(256 / C) signed digits of a canonical scalar at once, for a window width known at compile time: after unrolling
// every digit is cut out at a constant bit position (a shift, at most one funnel from the next limb, a mask) and the W results stay
// in registers - what the sort kernels use for the widths the library itself chooses (16 and 20); = msm_digit_step digit for digit
// (tests/host_harness).
template <int C>
LURK_HD void msm_digits_ct(const uint32_t (&s)[8], uint32_t (&d)[(256 + C - 1) / C]) {
    constexpr int W = (256 + C - 1) / C;
    constexpr uint32_t half = 1u << (C - 1), mask = (1u << C) - 1u;
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const int bit = w * C, limb = bit >> 5, sh = bit & 31;
        uint32_t raw = s[limb] >> sh;
        if (sh + C > 32 && limb + 1 < 8) raw |= s[limb + 1] << (32 - sh);
        raw = (raw & mask) + carry;
        const bool neg = raw > half;
        carry = neg ? 1u : 0u;
        d[w] = neg ? (((1u << C) - raw) | MSM_SIGN) : raw;
    }
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

// Host Horner over the plane sums of G key spaces, v[g] = [S_g, P_g0 .. P_g(c-2)]: sum_g 2^(c g) (S_g + sum_k 2^k P_gk)
template <class P>
LURK_HD Xyzz<P> msm_planes_horner_windows(const Xyzz<P>* v, int G, int c) {
    Xyzz<P> acc = xyzz_identity<P>();
    for (int g = G - 1; g >= 0; g--) {
        const Xyzz<P>* vg = v + (size_t)g * c;
        for (int k = c - 1; k >= 0; k--) {
            acc = xyzz_dbl<P>(acc);
            if (k <= c - 2) xyzz_add<P>(acc, vg[1 + k]);
        }
        xyzz_add<P>(acc, vg[0]);
    }
    return acc;
}

}  // namespace lurk

// ipa.hip - the vector and key folds of the inner-product argument that opens CompressedSNARK's polynomial commitments on the
// Pasta cycle.
//
// Reference: EE1 / EE2 = nova::provider::ipa_pc::EvaluationEngine for Pallas / Vesta (/root/reference/src/proof/nova.rs:57-62),
// run by CompressedSNARK::prove (nova.rs:341-356).  One round of the published argument (arecibo ipa_pc.rs, un-vendored; restated
// in oracle/pyref.py, parity unpinned), n = current length:
//     c_L = <a_L, b_R>, c_R = <a_R, b_L>                        -> lurk_hip_inner_product_dev
//     L = commit(ck_R || ck_c, a_L || c_L), R = commit(ck_L || ck_c, a_R || c_R)   -> the MSM entry points over device bases
//     r = transcript(L, R)                                                           (host)
//     a' = r a_L + r^-1 a_R,  b' = r^-1 b_L + r b_R               -> lurk_hip_fold_halves_dev
//     ck' = [r^-1] ck_L + [r] ck_R                                -> lurk_hip_points_fold_halves_dev
// The key fold is the expensive part (n/2 double scalar multiplications with full-size scalars per round): one lane per output
// point runs a joint double-and-add over the two scalars (Shamir: L, R and L + R as affine addends), so every lane of the
// launch takes the same branch at every bit - the scalars are launch-wide - and finishes with its own inversion.
#include <memory>

#include <vector>

#include "common.hpp"
#include "curve.cuh"
#include "curve29.cuh"
#include "msm_core.cuh"
#include "keyfold_plan.hpp"

namespace lurk {

constexpr int IPA_BLOCK = 256;

template <class F>
__global__ __launch_bounds__(IPA_BLOCK) void inner_product_kernel(const Fe<F>* __restrict__ a, const Fe<F>* __restrict__ b, size_t n,
                                                                    Fe<F>* __restrict__ partial) {
    __shared__ uint4 raw[IPA_BLOCK * 2];
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(raw);
    Fe<F> acc = fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * IPA_BLOCK) acc = fe_add<F>(acc, fe_mul<F>(a[i], b[i]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = IPA_BLOCK / 2; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fe_add<F>(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

template <class F>
__global__ __launch_bounds__(IPA_BLOCK) void fill_kernel(Fe<F>* __restrict__ v, size_t n, Fe<F> x) {
    for (size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * IPA_BLOCK) v[i] = x;
}

// both cross inner products of a round in one pass: <a_lo, b_hi> and <a_hi, b_lo>, block partials [2 * block + {0, 1}]
template <class F>
__global__ __launch_bounds__(IPA_BLOCK) void cross_products_kernel(const Fe<F>* __restrict__ a, const Fe<F>* __restrict__ b, size_t h,
                                                                     Fe<F>* __restrict__ partial) {
    __shared__ uint4 raw[2 * 2 * (IPA_BLOCK / 64)];
    Fe<F>(*sh)[IPA_BLOCK / 64] = reinterpret_cast<Fe<F>(*)[IPA_BLOCK / 64]>(raw);
    Fe<F> l = fe_zero<F>(), r = fe_zero<F>();
    for (size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x; i < h; i += (size_t)gridDim.x * IPA_BLOCK) {
        const Fe<F> alo = a[i], ahi = a[i + h], blo = b[i], bhi = b[i + h];
        l = fe_add<F>(l, fe_mul<F>(alo, bhi));
        r = fe_add<F>(r, fe_mul<F>(ahi, blo));
    }
    for (int off = 32; off >= 1; off >>= 1) {
        Fe<F> ol, orr;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ol.l[k] = __shfl_down(l.l[k], off);
            orr.l[k] = __shfl_down(r.l[k], off);
        }
        l = fe_add<F>(l, ol);
        r = fe_add<F>(r, orr);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sh[0][wave] = l;
        sh[1][wave] = r;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        Fe<F> t = sh[threadIdx.x][0];
        for (int w = 1; w < IPA_BLOCK / 64; w++) t = fe_add<F>(t, sh[threadIdx.x][w]);
        partial[2 * blockIdx.x + threadIdx.x] = t;
    }
}

// v[i] <- s_lo v[i] + s_hi v[h + i], i < h
template <class F>
__global__ __launch_bounds__(IPA_BLOCK) void fold_halves_kernel(Fe<F>* v, size_t h, Fe<F> s_lo, Fe<F> s_hi) {
    for (size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x; i < h; i += (size_t)gridDim.x * IPA_BLOCK)
        v[i] = fe_add<F>(fe_mul<F>(s_lo, v[i]), fe_mul<F>(s_hi, v[h + i]));
}

// The rounds without a folded key.  The folded key of round k is  ck_k[p] = sum over the originals i = p mod m (m = n / 2^k)
// of coef_k[i] * ck[i], coef_k[i] = the product of the fold weights i met so far; so
//     <a_L, ck_k R> = sum_{i : (i mod m) >= m/2} (a[(i mod m) - m/2] * coef_k[i]) * ck[i]
//     <a_R, ck_k L> = sum_{i : (i mod m) <  m/2} (a[(i mod m) + m/2] * coef_k[i]) * ck[i]
// are ordinary commitments under the ORIGINAL key - the resident table key the witness commitments use - of two length-n
// vectors with n/2 non-zeros each (zero digits never enter the MSM's sort).  A round then costs two table-mode MSMs instead of
// m/2 full-size double scalar multiplications whose 255-step ladder is latency-bound however short the key has become.
template <class F>
__global__ __launch_bounds__(IPA_BLOCK) void ipa_round_scalars_kernel(const Fe<F>* __restrict__ a, size_t m, const Fe<F>* __restrict__ coef, size_t n,
                                                                        Fe<F>* __restrict__ out_l, Fe<F>* __restrict__ out_r) {
    const size_t h = m >> 1;
    for (size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * IPA_BLOCK) {
        const size_t p = i & (m - 1);
        const bool hi = p >= h;
        const Fe<F> v = fe_mul<F>(a[hi ? p - h : p + h], coef[i]);
        if (out_r == nullptr) {  // one merged vector: the supports of L (upper halves) and R (lower halves) are disjoint; the pair is
            out_l[i] = v;         // committed in one pass with bit log2(m / 2) of the index as the selector (lurk_hip_msm_ctx_submit_pair_dev)
            continue;
        }
        out_l[i] = hi ? v : fe_zero<F>();
        out_r[i] = hi ? fe_zero<F>() : v;
    }
}
// coef[i] <- coef[i] * (s_lo if (i mod m) < m/2 else s_hi): the key folds as ck' = [s_lo] ck_L + [s_hi] ck_R
template <class F>
__global__ __launch_bounds__(IPA_BLOCK) void ipa_coef_fold_kernel(Fe<F>* __restrict__ coef, size_t n, size_t m, Fe<F> s_lo, Fe<F> s_hi) {
    const size_t h = m >> 1;
    for (size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * IPA_BLOCK)
        coef[i] = fe_mul<F>(coef[i], (i & (m - 1)) >= h ? s_hi : s_lo);
}

struct Scalar256 {
    uint32_t w[8];  // canonical
};

// out[i] = [k_lo] P[i] + [k_hi] P[h + i]
template <class P>
__global__ __launch_bounds__(IPA_BLOCK) void points_fold_halves_kernel(const Affine<P>* __restrict__ pts, size_t h, Scalar256 k_lo, Scalar256 k_hi, int top_bit,
                                                                         Affine<P>* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x;
    if (i >= h) return;
    const Affine<P> L = pts[i], R = pts[h + i];
    Xyzz<P> t = xyzz_from_affine<P>(L);
    xyzz_add<P>(t, xyzz_from_affine<P>(R));
    const Affine<P> T = xyzz_to_affine<P>(t);  // L + R (the identity stays (0, 0): madd skips it)
    Xyzz<P> acc = xyzz_identity<P>();
    for (int bit = top_bit; bit >= 0; bit--) {
        acc = xyzz_dbl<P>(acc);
        const uint32_t bl = (k_lo.w[bit >> 5] >> (bit & 31)) & 1u, bh = (k_hi.w[bit >> 5] >> (bit & 31)) & 1u;  // launch-wide: no divergence
        if (bl & bh) xyzz_madd<P>(acc, T, false);
        else if (bl) xyzz_madd<P>(acc, L, false);
        else if (bh) xyzz_madd<P>(acc, R, false);
    }
    out[i] = xyzz_to_affine<P>(acc);
}

template <class F>
static Scalar256 canonical_scalar(const void* s32_mont) {
    Fe<F> s;
    memcpy(s.l, s32_mont, 32);
    s = fe_from_mont<F>(s);
    Scalar256 k;
    for (int i = 0; i < 8; i++) k.w[i] = s.l[i];
    return k;
}
static int top_bit_of(const Scalar256& a, const Scalar256& b) {
    for (int bit = 255; bit >= 0; bit--)
        if (((a.w[bit >> 5] | b.w[bit >> 5]) >> (bit & 31)) & 1u) return bit;
    return -1;
}

template <class F>
static void inner_product(const void* d_a, const void* d_b, size_t n, void* out32, hipStream_t s) {
    unsigned blocks = n ? div_up(n, IPA_BLOCK) : 1, cap = (unsigned)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    ArenaBuf partial_buf((size_t)blocks * 32, s);
    Fe<F>* partial = (Fe<F>*)partial_buf.p;
    {
        ProfScope ps("ipa_inner_product", s);
        hipLaunchKernelGGL((inner_product_kernel<F>), dim3(blocks), dim3(IPA_BLOCK), 0, s, (const Fe<F>*)d_a, (const Fe<F>*)d_b, n, partial);
    }
    hipError_t e = hipGetLastError();
    std::vector<uint64_t> host((size_t)blocks * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), partial, host.size() * 8, hipMemcpyDeviceToHost, s);
    LURK_HIP_CHECK(e);
    LURK_HIP_CHECK(hipStreamSynchronize(s));
    Fe<F> acc = fe_zero<F>();
    for (unsigned b = 0; b < blocks; b++) {
        Fe<F> x;
        memcpy(x.l, host.data() + (size_t)b * 4, 32);
        acc = fe_add<F>(acc, x);
    }
    memcpy(out32, acc.l, 32);
}

template <class F>
static void fold_halves(void* d_v, size_t len, const void* lo32, const void* hi32, hipStream_t s) {
    const size_t h = len / 2;
    Fe<F> a, b;
    memcpy(a.l, lo32, 32);
    memcpy(b.l, hi32, 32);
    unsigned blocks = div_up(h, IPA_BLOCK), cap = (unsigned)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    ProfScope ps("ipa_fold_halves", s);
    hipLaunchKernelGGL((fold_halves_kernel<F>), dim3(blocks), dim3(IPA_BLOCK), 0, s, (Fe<F>*)d_v, h, a, b);
    LURK_HIP_CHECK(hipGetLastError());
}

template <class F>
static void ipa_round_scalars(const void* d_a, size_t m, const void* d_coef, size_t n, void* d_l, void* d_r, hipStream_t s) {
    unsigned blocks = div_up(n, IPA_BLOCK), cap = (unsigned)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    ProfScope ps("ipa_round_scalars", s);
    hipLaunchKernelGGL((ipa_round_scalars_kernel<F>), dim3(blocks), dim3(IPA_BLOCK), 0, s, (const Fe<F>*)d_a, m, (const Fe<F>*)d_coef, n, (Fe<F>*)d_l,
                       (Fe<F>*)d_r);
    LURK_HIP_CHECK(hipGetLastError());
}
template <class F>
static void ipa_coef_fold(void* d_coef, size_t n, size_t m, const void* lo32, const void* hi32, hipStream_t s) {
    Fe<F> a, b;
    memcpy(a.l, lo32, 32);
    memcpy(b.l, hi32, 32);
    unsigned blocks = div_up(n, IPA_BLOCK), cap = (unsigned)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    ProfScope ps("ipa_coef_fold", s);
    hipLaunchKernelGGL((ipa_coef_fold_kernel<F>), dim3(blocks), dim3(IPA_BLOCK), 0, s, (Fe<F>*)d_coef, n, m, a, b);
    LURK_HIP_CHECK(hipGetLastError());
}

template <class P, class SF>
static void points_fold_halves(const void* d_pts, size_t len, const void* lo32, const void* hi32, void* d_out, hipStream_t s) {
    const size_t h = len / 2;
    const Scalar256 kl = canonical_scalar<SF>(lo32), kh = canonical_scalar<SF>(hi32);
    ProfScope ps("ipa_points_fold", s);
    hipLaunchKernelGGL((points_fold_halves_kernel<P>), dim3(div_up(h, IPA_BLOCK)), dim3(IPA_BLOCK), 0, s, (const Affine<P>*)d_pts, h, kl, kh, top_bit_of(kl, kh),
                       (Affine<P>*)d_out);
    LURK_HIP_CHECK(hipGetLastError());
}


// ---- the key folded by k rounds' weights at once --------------------------------------------------------------------------------
// After k rounds the folded key is  ck_k[p] = sum_{b < T} s_b * ck[b m + p],  T = 2^k, m = n / T, with ONE set of T weights s_b (the
// products of the rounds' fold weights) for every p: m small multi-scalar multiplications that share their scalars.  Under a window-table
// key the table already holds 2^(kb j) ck[i] (kb = the key's window width, j < Wt), so with s_b = sum_j 2^(kb j) chunk_{b,j} and every
// chunk split into U signed sub-digits (widths wd_u, offsets off_u)
//     ck_k[p] = sum_u 2^(off_u) * sum_{j, b} digit_{b,j,u} * table[j n + b m + p]
// - U independent bucket problems per output with Wt T terms each and digits of <= 2^(wd - 1) in magnitude, and only off_{U-1} < kb
// doublings at the end.  The digit pattern is the same for every p: the host sorts the (j, b) pairs of every sub-digit slot by digit
// magnitude ONCE (T <= 2^12 weights), a lane takes one (p, slot) and walks the list from the largest magnitude down - a running sum
// that gains the points of the current magnitude (mixed additions from the table, radix-2^29 layer) and is added into the total once
// per magnitude: Wt T mixed + 2^(wd - 1) full additions per lane, every lane of a wave in the same iteration of the same loop, the
// loads of a wave 64 consecutive table records.  When m U is too few lanes to fill the chip a slot's window range is cut into groups.
struct KeyFoldDev {
    const uint32_t* ord;
    const uint32_t* dstart;
    const uint32_t* ord_base;  // U * groups entries (as 32-bit offsets)
    const uint32_t* jlo;       // groups entries
    uint32_t m, log_t, groups, maxmag, lanes_per_slot;
    size_t stride;             // the key's points: table[j * stride + i]
};

// lane (p, slot = u * groups + g): partial[slot * m + p] = sum over the slot's (j, b) of digit * table[j stride + b m + p]
template <class P>
__global__ __launch_bounds__(IPA_BLOCK) void key_fold_slots_kernel(const Affine<P>* __restrict__ table, KeyFoldDev d, uint32_t nslots, Xyzz<P>* __restrict__ partial) {
    const size_t gid = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x;
    if (gid >= (size_t)nslots * d.m) return;
    const uint32_t p = (uint32_t)(gid % d.m), slot = (uint32_t)(gid / d.m);
    const uint32_t* ord = d.ord + d.ord_base[slot];
    const uint32_t* ds = d.dstart + (size_t)slot * (d.maxmag + 1);
    const Affine<P>* base = table + (size_t)d.jlo[slot % d.groups] * d.stride + p;
    const uint32_t tmask = (1u << d.log_t) - 1u;
    Xyzz29<P> run, tot;
    run.x = run.y = run.zz = run.zzz = f29_zero<P>();
    tot = run;
    bool run_id = true, tot_id = true;
    uint32_t t = ds[0];
#pragma unroll 1
    for (uint32_t k = 0; k < d.maxmag; k++) {  // magnitude maxmag - k
        const uint32_t end = ds[k + 1];
#pragma unroll 1
        for (; t < end; t++) {
            const uint32_t e = ord[t], jb = e & 0x7fffffffu;
            const Affine<P> q = base[(size_t)(jb >> d.log_t) * d.stride + (size_t)(jb & tmask) * d.m];
            xyzz29_madd<P>(run, run_id, q, (e & 0x80000000u) != 0);
        }
        xyzz29_add<P>(tot, tot_id, run, run_id);
    }
    partial[gid] = xyzz29_to_xyzz<P>(tot, tot_id);
}

// lane p: out[p] = sum_u 2^(off_u) sum_g partial[(u, g)][p], as an affine point
template <class P>
__global__ __launch_bounds__(IPA_BLOCK) void key_fold_combine_kernel(const Xyzz<P>* __restrict__ partial, uint32_t m, int U, int groups, int off1, int off2,
                                                                       int off3, int off4, int off5, int off6, int off7, Affine<P>* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * IPA_BLOCK + threadIdx.x;
    if (p >= m) return;
    const int off[8] = {0, off1, off2, off3, off4, off5, off6, off7};
    Xyzz<P> acc = xyzz_identity<P>();
#pragma unroll 1
    for (int u = U - 1; u >= 0; u--) {
        if (u != U - 1) acc = xyzz_dbl_n<P>(acc, off[u + 1] - off[u]);
#pragma unroll 1
        for (int g = 0; g < groups; g++) xyzz_add<P>(acc, partial[((size_t)u * groups + g) * m + p]);
    }
    out[p] = xyzz_to_affine<P>(acc);
}

// weights: T Montgomery scalars on the host; out: m = n / T affine points on the device
template <class P, class F>
static void key_fold(const MsmTableView& v, size_t n, const void* weights32_mont, size_t T, void* d_out, hipStream_t s) {
    LURK_REQUIRE(T >= 1 && (T & (T - 1)) == 0 && T <= (size_t)KEYFOLD_MAX_T, "the number of fold weights must be a power of two, at most 2^12");
    LURK_REQUIRE(n >= T && n % T == 0 && n <= v.npoints, "the folded length must be a multiple of the weights and at most the key's points");
    const size_t m = n / T;
    LURK_REQUIRE(m < ((size_t)1 << 31), "too many output points");
    std::vector<uint64_t> canon(4 * T);
    for (size_t b = 0; b < T; b++) {
        Fe<F> x;
        memcpy(x.l, (const char*)weights32_mont + 32 * b, 32);
        x = fe_from_mont<F>(x);
        memcpy(canon.data() + 4 * b, x.l, 32);
    }
    LURK_REQUIRE(v.form == LURK_MSM_FORM_TABLE, "the key fold needs the window-table form of the key (LURK_MSM_FLAG_PRECOMPUTE, more than 2^16 points or a window-bit override)");
    const int kb = v.window_bits, Wt = v.windows;
    const KeyFoldPlan pl = keyfold_plan(canon.data(), T, m, kb, Wt);
    LURK_REQUIRE(pl.ok, "a fold weight does not fit the key's windows");
    int log_t = 0;
    while (((size_t)1 << log_t) < T) log_t++;
    const uint32_t nslots = (uint32_t)(pl.U * pl.groups);
    using Scratch = ArenaBuf;  // the stream's scratch arena (common.hpp)
    std::vector<uint32_t> base32(nslots), jlo32(pl.groups);
    for (uint32_t i = 0; i < nslots; i++) {
        LURK_REQUIRE(pl.ord_base[i] < ((size_t)1 << 32), "fold plan too large");
        base32[i] = (uint32_t)pl.ord_base[i];
    }
    for (int g = 0; g < pl.groups; g++) jlo32[g] = (uint32_t)pl.jlo[g];
    Scratch d_ord(pl.ord.size() * 4, s), d_ds(pl.dstart.size() * 4, s), d_base(nslots * 4, s), d_jlo(pl.groups * 4, s), d_part((size_t)nslots * m * sizeof(Xyzz<P>), s);
    if (!pl.ord.empty()) LURK_HIP_CHECK(hipMemcpyAsync(d_ord.p, pl.ord.data(), pl.ord.size() * 4, hipMemcpyHostToDevice, s));
    LURK_HIP_CHECK(hipMemcpyAsync(d_ds.p, pl.dstart.data(), pl.dstart.size() * 4, hipMemcpyHostToDevice, s));
    LURK_HIP_CHECK(hipMemcpyAsync(d_base.p, base32.data(), nslots * 4, hipMemcpyHostToDevice, s));
    LURK_HIP_CHECK(hipMemcpyAsync(d_jlo.p, jlo32.data(), pl.groups * 4, hipMemcpyHostToDevice, s));
    KeyFoldDev d;
    d.ord = (const uint32_t*)d_ord.p;
    d.dstart = (const uint32_t*)d_ds.p;
    d.ord_base = (const uint32_t*)d_base.p;
    d.jlo = (const uint32_t*)d_jlo.p;
    d.m = (uint32_t)m;
    d.log_t = (uint32_t)log_t;
    d.groups = (uint32_t)pl.groups;
    d.maxmag = (uint32_t)pl.maxmag;
    d.lanes_per_slot = (uint32_t)m;
    d.stride = v.npoints;
    {
        ProfScope ps("key_fold", s);
        hipLaunchKernelGGL((key_fold_slots_kernel<P>), dim3(div_up((size_t)nslots * m, IPA_BLOCK)), dim3(IPA_BLOCK), 0, s, (const Affine<P>*)v.table, d, nslots,
                           (Xyzz<P>*)d_part.p);
        hipLaunchKernelGGL((key_fold_combine_kernel<P>), dim3(div_up(m, IPA_BLOCK)), dim3(IPA_BLOCK), 0, s, (const Xyzz<P>*)d_part.p, (uint32_t)m, pl.U, pl.groups,
                           pl.off[1], pl.off[2], pl.off[3], pl.off[4], pl.off[5], pl.off[6], pl.off[7], (Affine<P>*)d_out);
        LURK_HIP_CHECK(hipGetLastError());
    }
    LURK_HIP_CHECK(hipStreamSynchronize(s));  // the plan's host vectors are read by the copies above
}

// ---- the whole argument under a resident key (arecibo ipa_pc::InnerProductArgument::prove, /root/reference/src/proof/nova.rs:57-62) ----
// The round loop of lurk_beta_amd/ipa.py: _prove_resident_key as host code of the library: per round one launch for the composed scalars,
// one pair commitment (window-table key) or two commitments in flight (plain / small key), one launch + one copy for both cross inner
// products, the two host scalar multiples of the extra base while the commitment runs, the transcript's challenge (a callback), one
// host inversion and three fold launches.  In Python the glue around the same calls cost 0.4-0.5 ms per round with the device idle.
// Rounds under a long key cost a full-size commitment each however short the vectors have become.  So after IPA_FOLD_ROUNDS rounds under
// a window-table key of >= 2^ipa_fold_min_log() points the key IS folded - once, by the product weights of those rounds (key_fold: about
// two commitments' worth of additions) - into a window-table key of n / 2^IPA_FOLD_ROUNDS points, and the argument continues under that
// one (which folds again if it is still long).  2^20: 4 rounds at 1.9 ms + the fold instead of 20 rounds at 1.9 ms.
constexpr int IPA_FOLD_ROUNDS_DEFAULT = 4;
static int ipa_fold_rounds() {
    const char* e = getenv("LURK_IPA_FOLD_ROUNDS");  // rounds under the long key before it is folded (1 .. 12); read per call (tests move it)
    const int v = e ? atoi(e) : IPA_FOLD_ROUNDS_DEFAULT;
    return v < 1 ? 1 : v > 12 ? 12 : v;
}
static int ipa_fold_min_log() {
    const char* e = getenv("LURK_IPA_FOLD_MIN_LOG");  // 0 = never fold the key (the round-3 form); default 2^18 points; read per call (tests move it)
    return e ? atoi(e) : 18;
}

template <class P, class F>
static void ipa_prove_resident(lurk_hip_msm_ctx* key, int curve, int field_id, void* d_a, void* d_b, size_t n0, const void* ck_c_jac96,
                               lurk_hip_ipa_challenge_fn challenge, void* user, uint64_t* out_l, uint64_t* out_r, void* out_a_hat32,
                               void* out_ck_hat64, hipStream_t s, int round0 = 0) {
    auto ok = [](int rc) { if (rc != 0) throw HipFailure{rc, lurk_hip_last_error()}; };
    int kc = 0, kbits = 0, ktable = 0;
    size_t kn = 0;
    ok(lurk_hip_msm_ctx_info(key, &kc, &kn, &kbits, nullptr));
    ok(lurk_hip_msm_ctx_form(key, &ktable));
    LURK_REQUIRE(kc == curve, "the key is over another curve");
    LURK_REQUIRE(kn >= n0, "the key has fewer points than the vectors have elements");
    const bool pairs = ktable == LURK_MSM_FORM_TABLE;  // the window-table form commits L and R (disjoint supports) in one pass
    stream_pool_retain();
    // stream-ordered scratch (the pool keeps it between calls: a proof opens several of these arguments)
    using Scratch = ArenaBuf;  // the stream's scratch arena (common.hpp)
    const unsigned max_blocks = (unsigned)num_cus() * 8;
    Scratch coef(n0 * 32, s), dl(n0 * 32, s), dr(pairs ? 0 : n0 * 32, s), partial((size_t)2 * max_blocks * 32, s);
    {
        unsigned blocks = div_up(n0, IPA_BLOCK);
        if (blocks > max_blocks) blocks = max_blocks;
        hipLaunchKernelGGL((fill_kernel<F>), dim3(blocks), dim3(IPA_BLOCK), 0, s, (Fe<F>*)coef.p, n0, fe_one<F>());
        LURK_HIP_CHECK(hipGetLastError());
    }
    std::vector<uint64_t> host_partial_buf((size_t)2 * max_blocks * 4);  // <= 128 KiB per round: a pageable copy is a few microseconds more
    uint64_t* host_partial = host_partial_buf.data();
    size_t m = n0;
    int j = 0;
    const int fold_log = ipa_fold_min_log();
    const int IPA_FOLD_ROUNDS = ipa_fold_rounds();
    const bool will_fold = pairs && fold_log > 0 && n0 >= ((size_t)1 << fold_log) && n0 >= ((size_t)1 << (IPA_FOLD_ROUNDS + 1));
    while (m > 1) {
        if (will_fold && j == IPA_FOLD_ROUNDS) {
            const size_t T = (size_t)1 << IPA_FOLD_ROUNDS;  // coef[i] depends on i / m only: the weight of block b is coef[b m]
            std::vector<uint64_t> w(4 * T);
            LURK_HIP_CHECK(hipMemcpy2DAsync(w.data(), 32, coef.p, m * 32, 32, T, hipMemcpyDeviceToHost, s));
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            Scratch folded(m * sizeof(Affine<P>), s);
            key_fold<P, F>(msm_ctx_table_view(key), n0, w.data(), T, folded.p, s);
            FoldedKeyLease key2 = msm_ctx_folded_child(key, folded.p, m, s);  // a window-table key (16-bit windows) that belongs to `key`
            ipa_prove_resident<P, F>(key2.ctx, curve, field_id, d_a, d_b, m, ck_c_jac96, challenge, user, out_l + (size_t)12 * j, out_r + (size_t)12 * j, out_a_hat32,
                                     out_ck_hat64, s, round0 + j);
            return;
        }
        const size_t h = m / 2;
        int log_h = 0;
        while (((size_t)1 << log_h) < h) log_h++;
        ipa_round_scalars<F>(d_a, m, coef.p, n0, dl.p, pairs ? nullptr : dr.p, s);
        // whatever fails between here and the wait, the slots are drained before the call returns: the key stays usable
        struct Drain {
            lurk_hip_msm_ctx* key;
            bool pairs;
            int pending = 0;
            ~Drain() {
                uint64_t sink[24];
                if (pending >= 1) (void)(pairs ? lurk_hip_msm_ctx_wait_pair(key, 0, sink, sink + 12) : lurk_hip_msm_ctx_wait(key, 0, sink));
                if (pending >= 2) (void)lurk_hip_msm_ctx_wait(key, 1, sink);
            }
        } drain{key, pairs};
        if (pairs) {
            ok(lurk_hip_msm_ctx_submit_pair_dev(key, 0, dl.p, n0, 1, (void*)s, log_h));  // bit log2(m / 2) of the index set: L's support
            drain.pending = 1;
        } else {
            ok(lurk_hip_msm_ctx_submit_dev_mode(key, 0, dl.p, n0, 1, (void*)s, LURK_MSM_SUBMIT_FOREGROUND));
            drain.pending = 1;
            ok(lurk_hip_msm_ctx_submit_dev_mode(key, 1, dr.p, n0, 1, (void*)s, LURK_MSM_SUBMIT_FOREGROUND));
            drain.pending = 2;
        }
        unsigned blocks = div_up(h, IPA_BLOCK);
        if (blocks > max_blocks) blocks = max_blocks;
        hipLaunchKernelGGL((cross_products_kernel<F>), dim3(blocks), dim3(IPA_BLOCK), 0, s, (const Fe<F>*)d_a, (const Fe<F>*)d_b, h, (Fe<F>*)partial.p);
        LURK_HIP_CHECK(hipGetLastError());
        LURK_HIP_CHECK(hipMemcpyAsync(host_partial, partial.p, (size_t)2 * blocks * 32, hipMemcpyDeviceToHost, s));
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        Fe<F> cl = fe_zero<F>(), cr = fe_zero<F>();
        for (unsigned b = 0; b < blocks; b++) {
            Fe<F> x, y;
            memcpy(x.l, host_partial + (size_t)(2 * b) * 4, 32);
            memcpy(y.l, host_partial + (size_t)(2 * b + 1) * 4, 32);
            cl = fe_add<F>(cl, x);
            cr = fe_add<F>(cr, y);
        }
        uint64_t two[24], c_l[12], c_r[12];
        uint64_t* L = out_l + (size_t)12 * j;
        uint64_t* R = out_r + (size_t)12 * j;
        uint64_t tl[12], tr[12];
        ok(lurk_hip_point_mul(curve, tl, ck_c_jac96, cl.l, 1));  // two 255-bit host scalar multiples, under the commitment
        ok(lurk_hip_point_mul(curve, tr, ck_c_jac96, cr.l, 1));
        if (pairs) {
            // the commitments arrive as XYZZ points, take the multiples of the extra base and leave normalised through ONE inversion
            drain.pending = 0;
            Xyzz<P> xl, xr;
            msm_ctx_wait_pair_xyzz(key, 0, &xr, &xl);
            Jacobian<P> jl, jr;
            memcpy(&jl, tl, 96);
            memcpy(&jr, tr, 96);
            xyzz_add<P>(xl, xyzz_from_jacobian<P>(jl));
            xyzz_add<P>(xr, xyzz_from_jacobian<P>(jr));
            Affine<P> al, ar;
            xyzz_pair_to_affine<P>(xl, xr, al, ar);
            jl = jacobian_from_affine<P>(al);
            jr = jacobian_from_affine<P>(ar);
            memcpy(L, &jl, 96);
            memcpy(R, &jr, 96);
        } else {
            drain.pending = 0;
            const int rc0 = lurk_hip_msm_ctx_wait(key, 0, c_l), rc1 = lurk_hip_msm_ctx_wait(key, 1, c_r);
            ok(rc0);
            ok(rc1);
        }
        if (!pairs) {
            memcpy(two, c_l, 96);
            memcpy(two + 12, tl, 96);
            ok(lurk_hip_point_sum(curve, L, two, 2));
            memcpy(two, c_r, 96);
            memcpy(two + 12, tr, 96);
            ok(lurk_hip_point_sum(curve, R, two, 2));
        }
        uint64_t r_can[4] = {0, 0, 0, 0};
        LURK_REQUIRE(challenge(user, round0 + j, L, R, r_can) == 0, "the transcript callback failed");
        Fe<F> r;
        memcpy(r.l, r_can, 32);
        LURK_REQUIRE(!fe_canonical_ge_mod<F>(r.l), "the challenge is not reduced modulo the group order");
        r = fe_to_mont<F>(r);
        LURK_REQUIRE(!fe_is_zero<F>(r), "zero challenge");
        const Fe<F> ri = fe_inv<F>(r);
        fold_halves<F>(d_a, m, r.l, ri.l, s);
        fold_halves<F>(d_b, m, ri.l, r.l, s);
        ipa_coef_fold<F>(coef.p, n0, m, ri.l, r.l, s);
        m = h;
        j++;
    }
    // the final key element is the commitment of the coefficient vector (the verifier's s vector)
    uint64_t ck_hat[12];
    ok(lurk_hip_msm_ctx_run_dev(key, ck_hat, coef.p, n0, 1, (void*)s));
    Jacobian<P> jp;
    memcpy(&jp, ck_hat, 96);
    const Affine<P> aff = xyzz_to_affine<P>(xyzz_from_jacobian<P>(jp));  // Montgomery coordinates, (0, 0) for the identity
    memcpy(out_ck_hat64, &aff, 64);
    Fe<F> a0;
    LURK_HIP_CHECK(hipMemcpyAsync(a0.l, d_a, 32, hipMemcpyDeviceToHost, s));
    LURK_HIP_CHECK(hipStreamSynchronize(s));
    a0 = fe_from_mont<F>(a0);
    memcpy(out_a_hat32, a0.l, 32);
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_inner_product_dev(int field_id, const void* d_a, const void* d_b, size_t n, void* out32_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(out32_mont && (n == 0 || (d_a && d_b)), "null argument");
        if (field_id == 0) inner_product<PallasFp>(d_a, d_b, n, out32_mont, (hipStream_t)stream);
        else if (field_id == 1) inner_product<PallasFq>(d_a, d_b, n, out32_mont, (hipStream_t)stream);
        else inner_product<Bn254Fr>(d_a, d_b, n, out32_mont, (hipStream_t)stream);
    });
}

int lurk_hip_fold_halves_dev(int field_id, void* d_v, size_t len, const void* s_lo32_mont, const void* s_hi32_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(len >= 2 && len % 2 == 0, "length must be even and >= 2");
        LURK_REQUIRE(d_v && s_lo32_mont && s_hi32_mont, "null argument");
        if (field_id == 0) fold_halves<PallasFp>(d_v, len, s_lo32_mont, s_hi32_mont, (hipStream_t)stream);
        else if (field_id == 1) fold_halves<PallasFq>(d_v, len, s_lo32_mont, s_hi32_mont, (hipStream_t)stream);
        else fold_halves<Bn254Fr>(d_v, len, s_lo32_mont, s_hi32_mont, (hipStream_t)stream);
    });
}

int lurk_hip_ipa_round_scalars_dev(int field_id, const void* d_a, size_t m, const void* d_coef, size_t n, void* d_out_l, void* d_out_r, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(m >= 2 && (m & (m - 1)) == 0 && n >= m && n % m == 0, "m must be a power of two >= 2 that divides n");
        LURK_REQUIRE(d_a && d_coef && d_out_l, "null argument");  // d_out_r == NULL: the merged form
        if (field_id == 0) ipa_round_scalars<PallasFp>(d_a, m, d_coef, n, d_out_l, d_out_r, (hipStream_t)stream);
        else if (field_id == 1) ipa_round_scalars<PallasFq>(d_a, m, d_coef, n, d_out_l, d_out_r, (hipStream_t)stream);
        else ipa_round_scalars<Bn254Fr>(d_a, m, d_coef, n, d_out_l, d_out_r, (hipStream_t)stream);
    });
}

int lurk_hip_ipa_coef_fold_dev(int field_id, void* d_coef, size_t n, size_t m, const void* s_lo32_mont, const void* s_hi32_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(m >= 2 && (m & (m - 1)) == 0 && n >= m && n % m == 0, "m must be a power of two >= 2 that divides n");
        LURK_REQUIRE(d_coef && s_lo32_mont && s_hi32_mont, "null argument");
        if (field_id == 0) ipa_coef_fold<PallasFp>(d_coef, n, m, s_lo32_mont, s_hi32_mont, (hipStream_t)stream);
        else if (field_id == 1) ipa_coef_fold<PallasFq>(d_coef, n, m, s_lo32_mont, s_hi32_mont, (hipStream_t)stream);
        else ipa_coef_fold<Bn254Fr>(d_coef, n, m, s_lo32_mont, s_hi32_mont, (hipStream_t)stream);
    });
}

int lurk_hip_ipa_prove_dev(lurk_hip_msm_ctx* key, void* d_a, void* d_b, size_t n, const void* ck_c_jacobian96, lurk_hip_ipa_challenge_fn challenge,
                           void* user, void* out_l_jacobian96, void* out_r_jacobian96, void* out_a_hat32, void* out_ck_hat_affine64, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(key && d_a && d_b && ck_c_jacobian96 && challenge && out_a_hat32 && out_ck_hat_affine64, "null argument");
        LURK_REQUIRE(n >= 1 && (n & (n - 1)) == 0, "the vector length must be a power of two");
        LURK_REQUIRE(n == 1 || (out_l_jacobian96 && out_r_jacobian96), "null argument");
        int curve = 0, bits = 0, device = 0;
        size_t points = 0;
        if (lurk_hip_msm_ctx_info(key, &curve, &points, &bits, nullptr) != 0 || lurk_hip_msm_ctx_device(key, &device) != 0)
            throw HipFailure{LURK_HIP_ERR_INVALID_ARG, lurk_hip_last_error()};
        DeviceGuard dg(device);
        if (curve == LURK_CURVE_PALLAS)
            ipa_prove_resident<PallasFp, PallasFq>(key, curve, LURK_FIELD_PALLAS_FQ, d_a, d_b, n, ck_c_jacobian96, challenge, user, (uint64_t*)out_l_jacobian96,
                                                   (uint64_t*)out_r_jacobian96, out_a_hat32, out_ck_hat_affine64, (hipStream_t)stream);
        else
            ipa_prove_resident<PallasFq, PallasFp>(key, curve, LURK_FIELD_PALLAS_FP, d_a, d_b, n, ck_c_jacobian96, challenge, user, (uint64_t*)out_l_jacobian96,
                                                   (uint64_t*)out_r_jacobian96, out_a_hat32, out_ck_hat_affine64, (hipStream_t)stream);
    });
}

int lurk_hip_msm_ctx_fold_key_dev(lurk_hip_msm_ctx* key, size_t n, const void* weights32_mont, size_t n_weights, void* d_out_affine64, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(key && weights32_mont && d_out_affine64, "null argument");
        const MsmTableView v = msm_ctx_table_view(key);
        DeviceGuard dg(v.device);
        if (v.curve == LURK_CURVE_PALLAS) key_fold<PallasFp, PallasFq>(v, n, weights32_mont, n_weights, d_out_affine64, (hipStream_t)stream);
        else key_fold<PallasFq, PallasFp>(v, n, weights32_mont, n_weights, d_out_affine64, (hipStream_t)stream);
    });
}

int lurk_hip_points_fold_halves_dev(int curve, const void* d_points_affine64, size_t len, const void* s_lo32_mont, const void* s_hi32_mont,
                                    void* d_out_affine64, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE(len >= 2 && len % 2 == 0, "length must be even and >= 2");
        LURK_REQUIRE(d_points_affine64 && d_out_affine64 && s_lo32_mont && s_hi32_mont, "null argument");
        if (curve == LURK_CURVE_PALLAS) points_fold_halves<PallasFp, PallasFq>(d_points_affine64, len, s_lo32_mont, s_hi32_mont, d_out_affine64, (hipStream_t)stream);
        else points_fold_halves<PallasFq, PallasFp>(d_points_affine64, len, s_lo32_mont, s_hi32_mont, d_out_affine64, (hipStream_t)stream);
    });
}
}

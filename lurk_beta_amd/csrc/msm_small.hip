// msm_small.hip - the small-commitment path of the Pedersen MSM: n <= 2^16 points under a RESIDENT key, one launch.
//
// Who needs it: the secondary-curve half of every folding step commits ~10^4-element vectors twice
// (/root/reference/src/proof/nova.rs:291-293 -> arecibo NIFS::prove on the secondary circuit), and SuperNova's small
// coprocessor circuits do the same (/root/reference/src/proof/supernova.rs:242-244).  Through the bucket pipeline of msm.hip such a
// commitment is ~45 dependent launches and a 2^15-bucket reduction: 0.45 ms of latency for 14 us worth of additions.
//
// Why not LDS-resident buckets here: a lane needs ~4.5 us for one dependent mixed addition (2 200 VALU instructions at one
// wave-instruction per 4+ cycles) and ~7 us for a full XYZZ addition, so what a small commitment costs is the DEPTH of its
// dependency chain, not its volume.  Buckets put three chains in series - the fullest bucket of a workgroup (8-12 additions for
// 512 buckets), the sum_b b*B_b reduction over the buckets (>= log2(512) = 9 full additions however it is arranged) and the
// combination across workgroups (8 more) - about 170 us before launch and copy overheads.  MI355X has 288 GB of HBM and a
// commitment key is fixed for a whole proof, so this path removes the buckets instead:
//
//   table[((i W + w) << (c-1)) + m - 1] = m * 2^(c w) * P_i        m = 1 .. 2^(c-1), w < W = ceil(256 / c), affine, 64 B each
//   sum_i k_i P_i = sum_{i,w} sign(d_iw) * table[i, w, |d_iw|]      d_iw = signed c-bit digits of k_i
//
// i.e. a commitment is a plain SUM of <= W n gathered affine points: no sort, no buckets, no bucket reduction, no doublings.
// c = 8 (256 KiB of table per point, 2.6 GB for the 10^4-point secondary key) up to 2^14 points, c = 6 (86 KiB per point,
// 5.6 GB at 2^16) above.  One kernel:
//   1. a lane owns one scalar's windows w = sub, sub + S, .. (S lanes per scalar, so that n W entries spread over 256 workgroups
//      x 4 waves): Montgomery -> canonical once, signed digits by one walk over the windows, gathered bases summed with the
//      radix-2^29 mixed addition of the big path (curve29.cuh);
//   2. wave: 6 xor-butterfly levels of full XYZZ additions over __shfl_xor; workgroup: 2 more levels through LDS;
//   3. the workgroup's point goes to global memory; the LAST workgroup to arrive (agent-scope counter) folds the <= 256
//      workgroup points 16:1 with 4 more butterfly levels and stores <= 16 points straight into pinned host memory;
//   4. host: sums them (0.4 us per addition against 7 on a GPU lane) and normalises.
// Depth for 2^13 scalars: 4 mixed + 12 full additions on the device (~110 us).  Algorithmic HBM bytes: 32 B scalar + W x 64 B
// gathered records per point (latency-bound: the roofline that matters is the dependency depth above).
#include "common.hpp"
#include "msm_core.cuh"
#include "curve29.cuh"

namespace lurk {

constexpr int SMALL_BLOCK = 256;       // 4 waves: one per SIMD of a CU
constexpr int SMALL_MAX_GROUPS = 256;  // one workgroup per CU
constexpr int SMALL_FINAL_LEVELS = 4;  // the last workgroup folds 16:1; the host adds what is left

// ---- table setup: every multiple m * B, m = 1 .. H, of the window bases B = 2^(c w) P_i (affine, from msm_precompute_kernel) ----
// One lane per (i, w): the multiples are carried in XYZZ form ((X, Y) parked in the table, ZZ / ZZZ / running product of the ZZZ in
// scratch), then ONE inversion and Montgomery's trick walk back over them - the scheme of msm_precompute_kernel.
template <class P>
__global__ __launch_bounds__(256) void small_multiples_kernel(const Affine<P>* __restrict__ wbases /*[w * n + i]*/, size_t n, int W, uint32_t H,
                                                                Affine<P>* __restrict__ table, Fe<P>* __restrict__ scratch, size_t id_lo,
                                                                size_t id_hi) {
    const size_t id = id_lo + (size_t)blockIdx.x * 256 + threadIdx.x;  // = i * W + w; this launch covers [id_lo, id_hi)
    if (id >= id_hi) return;
    const size_t i = id / W, w = id % W;
    const Affine<P> b = wbases[w * n + i];
    Affine<P>* out = table + id * H;
    if (affine_is_identity<P>(b)) {
        for (uint32_t m = 0; m < H; m++) out[m] = b;
        return;
    }
    Fe<P>* sc = scratch + (id - id_lo) * (size_t)H * 3;  // [m][0] = ZZ, [1] = ZZZ, [2] = product of ZZZ_1..m   (m = 1 .. H-1; index 0 unused)
    out[0] = b;
    Xyzz<P> p = xyzz_from_affine<P>(b);
    Fe<P> prod = fe_one<P>();
    for (uint32_t m = 1; m < H; m++) {  // p = (m + 1) B
        xyzz_madd<P>(p, b, false);      // m = 1: the equal-points case -> affine doubling
        out[m] = Affine<P>{p.x, p.y};
        prod = fe_mul<P>(prod, p.zzz);
        sc[m * 3 + 0] = p.zz;
        sc[m * 3 + 1] = p.zzz;
        sc[m * 3 + 2] = prod;
    }
    Fe<P> inv = fe_inv<P>(prod);
    for (uint32_t m = H - 1; m >= 1; m--) {
        const Fe<P> zzz = sc[m * 3 + 1];
        const Fe<P> zzz_inv = m > 1 ? fe_mul<P>(inv, sc[(m - 1) * 3 + 2]) : inv;
        inv = fe_mul<P>(inv, zzz);
        const Fe<P> t = fe_mul<P>(sc[m * 3 + 0], zzz_inv);  // ZZ / ZZZ = 1 / Z
        const Fe<P> zz_inv = fe_sqr<P>(t);
        Affine<P> q = out[m];
        q.x = fe_mul<P>(q.x, zz_inv);
        q.y = fe_mul<P>(q.y, zzz_inv);
        out[m] = q;
    }
}

// ---- the commitment ---------------------------------------------------------------------------------------------------------
// a point of the reduction trees: radix-2^29 XYZZ + the identity flag (160 B in LDS / global memory)
template <class P>
struct Pt29 {
    Xyzz29<P> p;
    uint32_t id, pad[3];
};

template <class P>
__device__ __forceinline__ void pt29_shfl_xor(const Xyzz29<P>& v, bool v_id, int mask, Xyzz29<P>& o, bool& o_id) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        o.x.l[k] = __shfl_xor(v.x.l[k], mask);
        o.y.l[k] = __shfl_xor(v.y.l[k], mask);
        o.zz.l[k] = __shfl_xor(v.zz.l[k], mask);
        o.zzz.l[k] = __shfl_xor(v.zzz.l[k], mask);
    }
    o_id = __shfl_xor((int)v_id, mask) != 0;
}

template <class P, class SF>
__global__ __launch_bounds__(SMALL_BLOCK) void msm_small_kernel(const uint4* __restrict__ scalars, size_t n, int is_mont, const Affine<P>* __restrict__ table,
                                                                  int c, int W, uint32_t S /*lanes per scalar*/, Pt29<P>* __restrict__ group_pts,
                                                                  uint32_t* __restrict__ counter, Xyzz<P>* __restrict__ out /*pinned host*/) {
    __shared__ Pt29<P> sh[SMALL_BLOCK / 64];
    __shared__ uint32_t sh_ticket;
    const uint32_t G = gridDim.x, T = G * SMALL_BLOCK;
    const uint32_t t = blockIdx.x * SMALL_BLOCK + threadIdx.x;
    const uint32_t slots = T / S, slot = t / S, sub = t % S;  // `slots` scalars are in flight per pass; lanes beyond slots * S idle
    const uint32_t K = ((uint32_t)W + S - 1) / S;             // own windows per scalar: w = sub + k S
    Xyzz29<P> acc;
    acc.x = acc.y = acc.zz = acc.zzz = f29_zero<P>();
    bool acc_id = true;
    // every lane of a wave reaches the addition of its k-th own window in the same iteration (the digit walk in between is
    // integer work of a few instructions per window): a loop over w with "is it mine" inside serialised the S window
    // classes of a wave, 34 us per addition instead of 5
    for (size_t i = slot; i < n && slot < slots; i += slots) {
        Fe<SF> s;
        {
            const uint4 lo = scalars[2 * i], hi = scalars[2 * i + 1];
            s.l[0] = lo.x; s.l[1] = lo.y; s.l[2] = lo.z; s.l[3] = lo.w;
            s.l[4] = hi.x; s.l[5] = hi.y; s.l[6] = hi.z; s.l[7] = hi.w;
        }
        if (is_mont) s = fe_from_mont<SF>(s);
        uint32_t carry = 0;
        uint32_t r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = s.l[k];
        for (uint32_t j = 0; j < sub; j++) (void)msm_digit_next(r, c, carry);  // the windows below the first own one
        const Affine<P>* row = table + ((i * (size_t)W) << (c - 1));
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t w = sub + k * S;
            // (past the top window the registers are empty and the carry is 0: the digits are 0)
            const uint32_t d = msm_digit_next(r, c, carry);
            for (uint32_t j = 1; j < S; j++) (void)msm_digit_next(r, c, carry);  // the other lanes' windows up to the next own one
            const uint32_t mag = d & ~MSM_SIGN;
            if (mag) {
                const Affine<P> q = row[((size_t)w << (c - 1)) + mag - 1];
                xyzz29_madd<P>(acc, acc_id, q, (d & MSM_SIGN) != 0);
            }
        }
    }
    // wave: xor butterfly of full additions (every lane ends with the wave's sum); workgroup: two levels through LDS
    Xyzz29<P> o;
    bool o_id;
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        pt29_shfl_xor<P>(acc, acc_id, off, o, o_id);
        xyzz29_add<P>(acc, acc_id, o, o_id);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sh[wave].p = acc;
        sh[wave].id = acc_id;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        acc = sh[threadIdx.x].p;
        acc_id = sh[threadIdx.x].id != 0;
        xyzz29_add<P>(acc, acc_id, sh[threadIdx.x + 2].p, sh[threadIdx.x + 2].id != 0);
        sh[threadIdx.x].p = acc;
        sh[threadIdx.x].id = acc_id;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = sh[0].p;
        acc_id = sh[0].id != 0;
        xyzz29_add<P>(acc, acc_id, sh[1].p, sh[1].id != 0);
        group_pts[blockIdx.x].p = acc;
        group_pts[blockIdx.x].id = acc_id;
        __threadfence();  // the point is visible device-wide before the ticket is taken
        sh_ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (sh_ticket != G - 1) return;
    // last workgroup: fold the G workgroup points 16:1 and hand <= 16 points to the host
    __threadfence();
    acc_id = true;
    if (threadIdx.x < G) {
        acc = group_pts[threadIdx.x].p;
        acc_id = group_pts[threadIdx.x].id != 0;
    }
#pragma unroll 1
    for (int off = 1; off < (1 << SMALL_FINAL_LEVELS); off <<= 1) {
        pt29_shfl_xor<P>(acc, acc_id, off, o, o_id);
        xyzz29_add<P>(acc, acc_id, o, o_id);
    }
    if ((threadIdx.x & ((1 << SMALL_FINAL_LEVELS) - 1)) == 0 && threadIdx.x < G) out[threadIdx.x >> SMALL_FINAL_LEVELS] = xyzz29_to_xyzz<P>(acc, acc_id);
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this slot
    __threadfence_system();
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
int msm_small_window_bits(size_t n) { return n <= ((size_t)1 << 14) ? 8 : 6; }
size_t msm_small_table_entries(size_t n, int c) { return (n * (size_t)msm_num_windows(c)) << (c - 1); }
unsigned msm_small_groups(size_t n, int c) {
    const size_t entries = n * (size_t)msm_num_windows(c);
    size_t g = (entries + SMALL_BLOCK - 1) / SMALL_BLOCK;
    if (g < 1) g = 1;
    if (g > SMALL_MAX_GROUPS) g = SMALL_MAX_GROUPS;
    return (unsigned)g;
}
size_t msm_small_scratch_bytes() { return (size_t)1 << 30; }
unsigned msm_small_out_points(unsigned groups) { return (groups + (1u << SMALL_FINAL_LEVELS) - 1) >> SMALL_FINAL_LEVELS; }

// wbases: the per-window bases [w * n + i] = 2^(c w) P_i (msm_precompute_kernel's output); table: n * W * 2^(c-1) records
template <class P>
void msm_small_build_table(const Affine<P>* wbases, size_t n, int c, Affine<P>* table, hipStream_t s) {
    if (n == 0) return;
    const int W = msm_num_windows(c);
    const uint32_t H = 1u << (c - 1);
    // (i, w) pairs in chunks: the ZZ / ZZZ / running-product scratch of a chunk is 96 B per table entry, bounded to
    // MSM_SMALL_SCRATCH_BYTES whatever the key size (all at once it was 1.5x the table itself: 6.4 GB at 2^14 points)
    const size_t ids = n * (size_t)W, per_id = (size_t)H * 3 * sizeof(Fe<P>);
    size_t chunk = msm_small_scratch_bytes() / per_id;
    chunk = chunk < 256 ? 256 : (chunk / 256) * 256;
    if (chunk > ids) chunk = ids;
    DevBuf scratch(chunk * per_id);
    for (size_t lo = 0; lo < ids; lo += chunk) {
        const size_t hi = lo + chunk < ids ? lo + chunk : ids;
        ProfScope ps("msm_small_table", s);
        hipLaunchKernelGGL((small_multiples_kernel<P>), dim3(div_up(hi - lo, 256)), dim3(256), 0, s, wbases, n, W, H, table, scratch.as<Fe<P>>(), lo, hi);
        LURK_HIP_CHECK(hipGetLastError());
    }
    LURK_HIP_CHECK(hipStreamSynchronize(s));  // the scratch buffer is released here
}

// counter must be zero before the first launch (the kernel leaves it zero); out: pinned host memory, msm_small_out_points() records
size_t msm_small_group_bytes() { return (size_t)SMALL_MAX_GROUPS * 160; }
template <class P, class SF>
void msm_small_launch(const void* d_scalars, size_t n, int is_mont, const Affine<P>* table, int c, void* group_pts, uint32_t* counter, Xyzz<P>* out,
                      hipStream_t s) {
    static_assert(sizeof(Pt29<P>) == 160, "tree point record");
    const int W = msm_num_windows(c);
    const unsigned G = msm_small_groups(n, c);
    const size_t T = (size_t)G * SMALL_BLOCK;
    uint32_t S = (uint32_t)(T / n);
    if (S < 1) S = 1;
    if (S > (uint32_t)W) S = (uint32_t)W;
    ProfScope ps("msm_small", s);
    hipLaunchKernelGGL((msm_small_kernel<P, SF>), dim3(G), dim3(SMALL_BLOCK), 0, s, (const uint4*)d_scalars, n, is_mont, table, c, W, S, (Pt29<P>*)group_pts, counter, out);
    LURK_HIP_CHECK(hipGetLastError());
}

#define LURK_SMALL_INSTANTIATE(P, SF)                                                                                      \
    template void msm_small_build_table<P>(const Affine<P>*, size_t, int, Affine<P>*, hipStream_t);                         \
    template void msm_small_launch<P, SF>(const void*, size_t, int, const Affine<P>*, int, void*, uint32_t*, Xyzz<P>*, hipStream_t);
LURK_SMALL_INSTANTIATE(PallasFp, PallasFq)
LURK_SMALL_INSTANTIATE(PallasFq, PallasFp)

}  // namespace lurk

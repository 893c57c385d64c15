// msm_bucket_direct.hip - stages 3-5 of the Pippenger pipeline (msm.hip) in ONE launch for commitments with few buckets.
//
// The general path plans tasks of <= S sorted entries (4 launches), accumulates them (1), and sums a bucket's partials (2): seven
// dependent launches and a round trip of the partials through memory.  That is the right shape when the accumulation is long; under a
// key of <= 2^16 points with 16-bit windows (32 768 buckets per key space, ~16 entries per bucket) the whole accumulation is ~10
// dependent additions deep and those stages cost 32 + 80 + 54 us plus their boundaries - a third of a 0.42 ms commitment, and the
// opening argument runs sixteen of these per proof under its folded key (ipa.hip).
//
// Here L = 1, 2 or 4 adjacent lanes own one bucket: each sums a contiguous share of the bucket's sorted entries (mixed additions on
// the radix-2^29 layer, msm_task_accumulate29_raw), the shares meet in log2 L xor-butterfly steps (xyzz29_add over __shfl_xor), lane 0
// stores the bucket.  No task list, no partials, no order: cnt and bucket_start from the sort are all it reads.  A bucket with more
// than L x DIRECT_CAP entries (equal scalars: a whole window in one bucket) goes on the big list and is summed by a workgroup.
#include "common.hpp"
#include "msm_core.cuh"
#include "curve29.cuh"

namespace lurk {

constexpr int DIRECT_BLOCK = 256;
constexpr uint32_t DIRECT_CAP = 64;  // entries per lane above which a bucket is handed to a workgroup

template <class P, int L>
__global__ __launch_bounds__(DIRECT_BLOCK) void msm_bucket_direct_kernel(const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                                           const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ cnt,
                                                                           uint32_t NB, Xyzz<P>* __restrict__ buckets, uint32_t* __restrict__ big_list,
                                                                           uint32_t* __restrict__ big_count) {
    __builtin_amdgcn_s_setprio(1);
    const size_t lane = (size_t)blockIdx.x * DIRECT_BLOCK + threadIdx.x;
    size_t key = lane / L;
    const uint32_t sub = (uint32_t)(lane % L);
    const bool live = key < NB;  // (a group past the last bucket still walks the butterfly: the shuffles need every lane)
    if (!live) key = NB - 1;
    uint32_t n = live ? cnt[key] : 0u;
    const uint32_t start = bucket_start[key];
    const bool big = n > (uint32_t)L * DIRECT_CAP;
    if (big) {
        if (sub == 0) big_list[atomicAdd(big_count, 1u)] = (uint32_t)key;
        n = 0;
    }
    const uint32_t per = (n + L - 1) / L;
    uint32_t first = start + sub * per, last = first + per;
    if (first > start + n) first = start + n;
    if (last > start + n) last = start + n;
    // (a one-record-ahead prefetch of the table gather changed nothing: 146 us either way for the opening argument's pair commitments.
    // All NB x L lanes are resident at once, two waves per SIMD sharing the multiplier - ~9 us per addition - so the launch lasts as
    // long as the fullest bucket's share: 17 additions where the mean is 8.  The planned-task stages balance better and pay for it
    // in launches; the two forms end up within 10 % of each other, this one ahead by the plan stage's 30 us.)
    Xyzz29<P> acc;
    bool acc_id;
    msm_task_accumulate29_raw<P>(sorted, first, last, table, acc, acc_id);
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) {
        Xyzz29<P> o;
        bool o_id;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            o.x.l[k] = __shfl_xor(acc.x.l[k], off);
            o.y.l[k] = __shfl_xor(acc.y.l[k], off);
            o.zz.l[k] = __shfl_xor(acc.zz.l[k], off);
            o.zzz.l[k] = __shfl_xor(acc.zzz.l[k], off);
        }
        o_id = __shfl_xor((int)acc_id, off) != 0;
        xyzz29_add<P>(acc, acc_id, o, o_id);
    }
    if (live && !big && sub == 0) buckets[key] = xyzz29_to_xyzz<P>(acc, acc_id);
}

// the listed buckets, one workgroup each: lanes stride the bucket's entries, then an LDS tree
template <class P>
__global__ __launch_bounds__(DIRECT_BLOCK) void msm_bucket_direct_big_kernel(const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                                               const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ cnt,
                                                                               Xyzz<P>* __restrict__ buckets, const uint32_t* __restrict__ big_list,
                                                                               const uint32_t* __restrict__ big_count) {
    __builtin_amdgcn_s_setprio(1);
    extern __shared__ uint4 lds_raw[];
    Xyzz<P>* sh = reinterpret_cast<Xyzz<P>*>(lds_raw);
    const uint32_t nbig = *big_count;
    for (uint32_t i = blockIdx.x; i < nbig; i += gridDim.x) {
        const uint32_t key = big_list[i], n = cnt[key], start = bucket_start[key];
        Xyzz29<P> a29;
        a29.x = a29.y = a29.zz = a29.zzz = f29_zero<P>();
        bool a_id = true;
        for (uint32_t j = threadIdx.x; j < n; j += DIRECT_BLOCK) {
            const uint32_t e = sorted[start + j];
            xyzz29_madd<P>(a29, a_id, table[e & 0x7fffffffu], (e & 0x80000000u) != 0);
        }
        Xyzz<P> acc = xyzz29_to_xyzz<P>(a29, a_id);
        const int t = threadIdx.x;
        sh[t] = acc;
        __syncthreads();
        for (int stride = DIRECT_BLOCK / 2; stride >= 1; stride >>= 1) {
            if (t < stride) {
                xyzz_add<P>(acc, sh[t + stride]);
                sh[t] = acc;
            }
            __syncthreads();
        }
        if (t == 0) buckets[key] = acc;
        __syncthreads();
    }
}

// lanes per bucket: the most (<= 4) that keeps NB x L within the 131 072 lanes two waves per SIMD hold and leaves a lane >= 4 entries
// (one key space of 16-bit windows is 32 768 buckets: 4 lanes; a pair's two key spaces: 2)
int msm_bucket_direct_lanes(size_t NB, size_t entries) {
    int L = 1;
    while (L < 4 && NB * (size_t)(2 * L) <= 131072 && entries / (NB * (size_t)(2 * L)) >= 4) L *= 2;
    return L;
}

template <class P>
void msm_launch_bucket_direct(const uint32_t* sorted, const Affine<P>* table, const uint32_t* bucket_start, const uint32_t* cnt, uint32_t NB, size_t entries,
                              Xyzz<P>* buckets, uint32_t* big_list, uint32_t* big_count, hipStream_t s) {
    const int L = msm_bucket_direct_lanes(NB, entries);
    const dim3 grid((unsigned)div_up((size_t)NB * L, DIRECT_BLOCK)), block(DIRECT_BLOCK);
    if (L == 1) hipLaunchKernelGGL((msm_bucket_direct_kernel<P, 1>), grid, block, 0, s, sorted, table, bucket_start, cnt, NB, buckets, big_list, big_count);
    else if (L == 2) hipLaunchKernelGGL((msm_bucket_direct_kernel<P, 2>), grid, block, 0, s, sorted, table, bucket_start, cnt, NB, buckets, big_list, big_count);
    else hipLaunchKernelGGL((msm_bucket_direct_kernel<P, 4>), grid, block, 0, s, sorted, table, bucket_start, cnt, NB, buckets, big_list, big_count);
    hipLaunchKernelGGL((msm_bucket_direct_big_kernel<P>), dim3(128), block, DIRECT_BLOCK * sizeof(Xyzz<P>), s, sorted, table, bucket_start, cnt, buckets, big_list,
                       big_count);
}
#define LURK_DIRECT_INSTANTIATE(P)                                                                                                                      \
    template void msm_launch_bucket_direct<P>(const uint32_t*, const Affine<P>*, const uint32_t*, const uint32_t*, uint32_t, size_t, Xyzz<P>*, uint32_t*, \
                                              uint32_t*, hipStream_t);
LURK_DIRECT_INSTANTIATE(PallasFp)
LURK_DIRECT_INSTANTIATE(PallasFq)

}  // namespace lurk

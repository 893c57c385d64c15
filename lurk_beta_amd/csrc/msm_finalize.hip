// msm_finalize.hip - stage 5 of the Pippenger pipeline (msm.hip): a bucket's task partials become the bucket.
//
// One lane per bucket sums its <= MSM_SMALL partials; fuller buckets go on a list for msm_big_bucket_kernel (msm.hip).  A lane's
// additions are DEPENDENT, so what the stage costs is nt - 1 times the latency of one general addition on a lane: round 5 moved them
// from the 8 x 32 group law (~20 us each with one or two waves per SIMD) to the radix-2^29 layer (xyzz_sum_via29: 6.4 us).  It
// matters where a commitment is short: the opening argument's rounds under its folded 65 536-point key (8-entry tasks, 2-5 partials
// per bucket) spent 69 us of their 420 here.
#include "common.hpp"
#include "msm_core.cuh"
#include "curve29.cuh"

namespace lurk {

constexpr int MSM_FIN_SMALL = 16;  // = MSM_SMALL of msm.hip (buckets with more partials are summed by a workgroup)

template <class P>
__global__ __launch_bounds__(256, 4) void msm_finalize_kernel(const Xyzz<P>* __restrict__ partials, const uint32_t* __restrict__ cnt,
                                                             const uint32_t* __restrict__ task_start,
                                                             const uint32_t* __restrict__ group_task_base, uint32_t NB,
                                                             Xyzz<P>* __restrict__ buckets, uint32_t* __restrict__ big_list,
                                                             uint32_t* __restrict__ big_count, uint32_t S) {
    __builtin_amdgcn_s_setprio(3);  // a latency-bound tail kernel: issue ahead of an accumulation sharing the SIMD
    const size_t key = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (key >= NB) return;
    const uint32_t g = (uint32_t)(key / MSM_GRP), b = (uint32_t)(key % MSM_GRP);
    const uint32_t nt = (cnt[key] + S - 1) / S;
    const uint32_t first = group_task_base[g] + task_start[(size_t)g * (MSM_GRP + 1) + b];
    if (nt > MSM_FIN_SMALL) {
        big_list[atomicAdd(big_count, 1u)] = (uint32_t)key;
        return;
    }
    buckets[key] = xyzz_sum_via29<P>(partials + first, nt);
}

template <class P>
void msm_launch_finalize(const Xyzz<P>* partials, const uint32_t* cnt, const uint32_t* task_start, const uint32_t* group_task_base, uint32_t NB,
                         Xyzz<P>* buckets, uint32_t* big_list, uint32_t* big_count, uint32_t S, hipStream_t s) {
    hipLaunchKernelGGL((msm_finalize_kernel<P>), dim3(div_up((size_t)NB, 256)), dim3(256), 0, s, partials, cnt, task_start, group_task_base, NB, buckets,
                       big_list, big_count, S);
}
template void msm_launch_finalize<PallasFp>(const Xyzz<PallasFp>*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, Xyzz<PallasFp>*, uint32_t*,
                                            uint32_t*, uint32_t, hipStream_t);
template void msm_launch_finalize<PallasFq>(const Xyzz<PallasFq>*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, Xyzz<PallasFq>*, uint32_t*,
                                            uint32_t*, uint32_t, hipStream_t);

}  // namespace lurk

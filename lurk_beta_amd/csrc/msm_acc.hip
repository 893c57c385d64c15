// msm_acc.hip - the dominant kernel of the MSM (bucket accumulation), in its own translation unit.
// The accumulator lives on the radix-2^29 layer (curve29.cuh: products without carry folds, lazy
// reduction); LURK_ACC_RADIX29=0 builds the 8 x 32-bit Montgomery version with the multiplier inlined
// (kept for A/B measurements).
#ifndef LURK_ACC_RADIX29
#define LURK_ACC_RADIX29 1
#endif
#if !LURK_ACC_RADIX29
#define LURK_MUL_FORCE_INLINE
#endif
#include "common.hpp"
#include "msm_core.cuh"
#include "curve29.cuh"

namespace lurk {

constexpr int MSM_ACC_BLOCK = 256;

template <class P>
__global__ __launch_bounds__(MSM_ACC_BLOCK) void msm_accumulate_kernel(const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                                         const uint2* __restrict__ task_info, const uint32_t* __restrict__ order,
                                                                         const uint32_t* __restrict__ group_task_base, int NG,
                                                                         Xyzz<P>* __restrict__ partials) {
    uint32_t i = blockIdx.x * MSM_ACC_BLOCK + threadIdx.x;
    if (i >= group_task_base[NG]) return;
    uint32_t t = order[i];
    uint2 ti = task_info[t];
#if LURK_ACC_RADIX29
    partials[t] = msm_task_accumulate29<P>(sorted, ti.x, ti.y, table);
#else
    partials[t] = msm_task_accumulate<P>(sorted, ti.x, ti.y, table);
#endif
}


template <class P>
void msm_launch_accumulate(const uint32_t* sorted, const Affine<P>* table, const uint2* task_info, const uint32_t* order,
                           const uint32_t* group_task_base, int NG, Xyzz<P>* partials, size_t nt, hipStream_t s) {
    hipLaunchKernelGGL((msm_accumulate_kernel<P>), dim3(div_up(nt, MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, sorted, table, task_info, order,
                       group_task_base, NG, partials);
}
template void msm_launch_accumulate<PallasFp>(const uint32_t*, const Affine<PallasFp>*, const uint2*, const uint32_t*, const uint32_t*, int,
                                              Xyzz<PallasFp>*, size_t, hipStream_t);
template void msm_launch_accumulate<PallasFq>(const uint32_t*, const Affine<PallasFq>*, const uint2*, const uint32_t*, const uint32_t*, int,
                                              Xyzz<PallasFq>*, size_t, hipStream_t);

}  // namespace lurk

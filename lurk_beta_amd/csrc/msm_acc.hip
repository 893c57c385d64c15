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
#include "msm_acc_task.cuh"

namespace lurk {

// One launch covers every task (the hardware dispatcher balances the workgroups).
template <class P>
__global__ __launch_bounds__(MSM_ACC_BLOCK) void msm_accumulate_kernel(const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                                         const uint2* __restrict__ task_info, const uint32_t* __restrict__ order,
                                                                         const uint32_t* __restrict__ group_task_base, int NG,
                                                                         Xyzz<P>* __restrict__ partials) {
#if !defined(LURK_ACC_NO_SETPRIO)
    __builtin_amdgcn_s_setprio(1);  // above a persistent (background) accumulation sharing the SIMD, below every short kernel (3)
#endif
    uint32_t i = blockIdx.x * MSM_ACC_BLOCK + threadIdx.x;
    if (i >= group_task_base[NG]) return;
    msm_accumulate_task<P>(i, sorted, table, task_info, order, partials);
}

template <class P>
void msm_launch_accumulate(const uint32_t* sorted, const Affine<P>* table, const uint2* task_info, const uint32_t* order,
                           const uint32_t* group_task_base, int NG, Xyzz<P>* partials, size_t nt, hipStream_t s) {
    hipLaunchKernelGGL((msm_accumulate_kernel<P>), dim3(div_up(nt, MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s, sorted, table, task_info, order,
                       group_task_base, NG, partials);
}
#define LURK_ACC_INSTANTIATE(P)                                                                                                             \
    template void msm_launch_accumulate<P>(const uint32_t*, const Affine<P>*, const uint2*, const uint32_t*, const uint32_t*, int, Xyzz<P>*, \
                                           size_t, hipStream_t);
LURK_ACC_INSTANTIATE(PallasFp)
LURK_ACC_INSTANTIATE(PallasFq)

}  // namespace lurk

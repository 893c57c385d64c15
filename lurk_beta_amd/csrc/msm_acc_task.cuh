// msm_acc_task.cuh - the body both forms of the bucket accumulation run per task (msm_acc.hip: the plain launch; msm_acc_persistent.hip:
// the persistent form).  The two translation units may be built with different flags (Makefile: ACC_FLAGS / PERSIST_FLAGS), never with
// different code: one definition.
#pragma once
#include "msm_core.cuh"
#include "curve29.cuh"

namespace lurk {

constexpr int MSM_ACC_BLOCK = 256;

#ifndef LURK_ACC_TASK_NOINLINE
#define LURK_ACC_TASK_NOINLINE 0
#endif
#if LURK_ACC_TASK_NOINLINE
#define LURK_ACC_TASK_ATTR __attribute__((noinline))
#else
#define LURK_ACC_TASK_ATTR __forceinline__
#endif
template <class P>
__device__ LURK_ACC_TASK_ATTR void msm_accumulate_task(uint32_t i, const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                    const uint2* __restrict__ task_info, const uint32_t* __restrict__ order,
                                                    Xyzz<P>* __restrict__ partials) {
    uint32_t t = order[i];
    uint2 ti = task_info[t];
#if LURK_ACC_RADIX29
    partials[t] = msm_task_accumulate29<P>(sorted, ti.x, ti.y, table);
#else
    partials[t] = msm_task_accumulate<P>(sorted, ti.x, ti.y, table);
#endif
}

}  // namespace lurk

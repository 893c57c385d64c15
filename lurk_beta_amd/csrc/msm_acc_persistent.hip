// msm_acc_persistent.hip - the persistent form of the bucket accumulation (commitments in flight; round 6: the accumulation of a
// LURK_MSM_SUBMIT_FOLLOW commitment), in its own translation unit so that it CAN be built for another register footprint than the
// plain launch of msm_acc.hip (Makefile: PERSIST_FLAGS / PERSIST_DEFS).  The SHIPPED build sets neither: the kernel takes ~176 VGPRs,
// the same code as the plain launch (msm_acc_task.cuh).  What has to run beside resident accumulations is sized for THAT footprint and
// checked on the compiler's own numbers (tests/test_cabi_exports.py::test_sort_kernels_fit_beside_a_resident_accumulation): four waves
// of a 1024-thread sort workgroup beside ONE 176-register wave, a 128-register tail wave beside TWO.  (A 144-register build - machine
// LICM off - lets the 56-register sort passes sit beside two; measured in round 4, +1.7 % in flight, -4 % for the folding step: not shipped.)
#ifndef LURK_ACC_RADIX29
#define LURK_ACC_RADIX29 1
#endif
#if !LURK_ACC_RADIX29
#define LURK_MUL_FORCE_INLINE
#endif
#include "common.hpp"
#include "msm_acc_task.cuh"

namespace lurk {

// (XCC, SE, CU) of the running wave as one index < 512 (HW_ID: CU_ID [11:8], SE_ID [14:13]; XCC_ID [3:0])
__device__ __forceinline__ uint32_t msm_cu_index() {
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
    return (xcc & 7u) * 64u + ((hw >> 13) & 3u) * 16u + ((hw >> 8) & 15u);
}

// Persistent form for commitments in flight: a fixed number of waves per SIMD (the launch grid), each wave pulls the next
// 64 tasks of the longest-first order from a global cursor.  The kernel then never holds more than its share of every
// SIMD's registers and wave slots, so the latency / HBM-bound kernels of the NEXT commitment (sort, plan, bucket
// reduction: other stream, higher priority) find room on every CU while this one keeps the integer VALU busy.
template <class P>
#ifndef LURK_PERSIST_MIN_BLOCKS
#define LURK_PERSIST_MIN_BLOCKS 1  // (HIP: minimum WAVES per SIMD) 4: the compiler holds the kernel to 128 registers (and spills the rest)
#endif
__global__ __launch_bounds__(MSM_ACC_BLOCK, LURK_PERSIST_MIN_BLOCKS) void msm_accumulate_persistent_kernel(const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                                                    const uint2* __restrict__ task_info,
                                                                                    const uint32_t* __restrict__ order,
                                                                                    const uint32_t* __restrict__ group_task_base, int NG,
                                                                                    Xyzz<P>* __restrict__ partials, uint32_t* __restrict__ cursor) {
    const uint32_t ntasks = group_task_base[NG];
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) atomicAdd(&cursor[MSM_PLACEMENT_BASE + msm_cu_index()], 1u);  // diagnostic: workgroups per CU
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64u);
        base = __builtin_amdgcn_readfirstlane(base);  // wave-uniform: the loop control stays scalar
        if (base >= ntasks) break;
        if (base + lane < ntasks) msm_accumulate_task<P>(base + lane, sorted, table, task_info, order, partials);
    }
}

// one workgroup of 4 waves per CU (one wave per SIMD); cursor must be zero when the kernel starts
template <class P>
void msm_launch_accumulate_persistent(const uint32_t* sorted, const Affine<P>* table, const uint2* task_info, const uint32_t* order,
                                      const uint32_t* group_task_base, int NG, Xyzz<P>* partials, uint32_t* cursor, hipStream_t s, unsigned wgs_per_cu) {
    // LURK_MSM_PERSIST_WGS workgroups per CU (1: one wave per SIMD, the form two of which are resident at once; 2: two waves per SIMD - a
    // single accumulation at the multiplier's full rate, for LURK_MSM_MAX_ACC=1)
    static const unsigned wgs = [] { const char* e = getenv("LURK_MSM_PERSIST_WGS"); int v = e ? atoi(e) : 1; return (unsigned)(v < 1 ? 1 : v > 4 ? 4 : v); }();
    // wgs_per_cu != 0: the caller's choice for this launch (a LURK_MSM_SUBMIT_FOLLOW commitment: msm.hip)
    hipLaunchKernelGGL((msm_accumulate_persistent_kernel<P>), dim3((unsigned)num_cus() * (wgs_per_cu ? wgs_per_cu : wgs)), dim3(MSM_ACC_BLOCK), 0, s, sorted, table, task_info, order,
                       group_task_base, NG, partials, cursor);
}
#define LURK_ACC_PERSISTENT_INSTANTIATE(P)                                                                                                    \
    template void msm_launch_accumulate_persistent<P>(const uint32_t*, const Affine<P>*, const uint2*, const uint32_t*, const uint32_t*, int, \
                                                      Xyzz<P>*, uint32_t*, hipStream_t, unsigned);
LURK_ACC_PERSISTENT_INSTANTIATE(PallasFp)
LURK_ACC_PERSISTENT_INSTANTIATE(PallasFq)

}  // namespace lurk

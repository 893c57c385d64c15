// msm_precompute.hip - the window table of a resident key, T[w n + i] = 2^(c w) P_i (one lane per point; msm_precompute.cuh).
#include "common.hpp"
#include "msm_core.cuh"
#include "msm_precompute.cuh"

namespace lurk {

size_t msm_precompute_scratch_bytes(size_t n, int W) { return W > 1 ? (size_t)(W - 1) * MSM_PRE_SLOTS * n * sizeof(F29<PallasFp>) : 0; }

template <class P>
__global__ __launch_bounds__(256) void msm_precompute_kernel(const Affine<P>* __restrict__ bases, size_t n, Affine<P>* __restrict__ table, int c,
                                                               int W, F29<P>* __restrict__ scratch) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    msm_precompute_point<P>(bases[i], i, n, c, W, table, scratch);
}

template <class P>
void msm_launch_precompute(const Affine<P>* bases, size_t n, Affine<P>* table, int c, int W, void* scratch, hipStream_t s) {
    static_assert(sizeof(F29<P>) == 36, "scratch record");
    if (n == 0) return;
    hipLaunchKernelGGL((msm_precompute_kernel<P>), dim3(div_up(n, 256)), dim3(256), 0, s, bases, n, table, c, W, (F29<P>*)scratch);
    LURK_HIP_CHECK(hipGetLastError());
}
template void msm_launch_precompute<PallasFp>(const Affine<PallasFp>*, size_t, Affine<PallasFp>*, int, int, void*, hipStream_t);
template void msm_launch_precompute<PallasFq>(const Affine<PallasFq>*, size_t, Affine<PallasFq>*, int, int, void*, hipStream_t);

}  // namespace lurk

// latency_probe.hip - latency / throughput of one XYZZ addition chain, multiplier called vs inlined.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#ifdef PROBE_INLINE
#define LURK_MUL_FORCE_INLINE
#endif
#include "../lurk_beta_amd/csrc/curve.cuh"
using namespace lurk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class P>
__global__ void k_addchain(const Affine<P>* pts, Xyzz<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Xyzz<P> acc = xyzz_from_affine<P>(pts[2 * i]);
    Xyzz<P> q = xyzz_dbl<P>(xyzz_from_affine<P>(pts[2 * i + 1]));
    for (int k = 0; k < iters; k++) xyzz_add<P>(acc, q);
    out[i] = acc;
}
template <class P>
__global__ void k_maddchain(const Affine<P>* pts, Xyzz<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Xyzz<P> acc = xyzz_from_affine<P>(pts[2 * i]);
    Affine<P> q = pts[2 * i + 1];
    for (int k = 0; k < iters; k++) xyzz_madd<P>(acc, q, (k & 1) != 0);
    out[i] = acc;
}
static double time_kernel(std::function<void()> f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 3; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    using P = PallasFp;
    const int maxthreads = 256 * 256 * 8;
    // points: use small multiples of G computed on host via the same header (host path)
    std::vector<Affine<P>> h(maxthreads * 2);
    Affine<P> g; g.x = fe_neg<P>(fe_one<P>()); g.y = fe_dbl<P>(fe_one<P>());
    Xyzz<P> cur = xyzz_from_affine<P>(g);
    for (int i = 0; i < 64; i++) { h[i] = xyzz_to_affine<P>(cur); xyzz_madd<P>(cur, g, false); }
    for (size_t i = 64; i < h.size(); i++) h[i] = h[i % 64 == (i / 64) % 64 ? (i + 1) % 64 : i % 64];
    Affine<P>* d_in; Xyzz<P>* d_out;
    CK(hipMalloc(&d_in, h.size() * sizeof(Affine<P>))); CK(hipMalloc(&d_out, maxthreads * sizeof(Xyzz<P>)));
    CK(hipMemcpy(d_in, h.data(), h.size() * sizeof(Affine<P>), hipMemcpyHostToDevice));
    const int iters = 64;
#ifdef PROBE_INLINE
    const char* tag = "inline";
#else
    const char* tag = "call";
#endif
    struct Cfg { int blocks, threads; const char* name; } cfgs[] = {{1, 64, "1 wave"}, {256, 64, "1 wave/CU"}, {1024, 64, "1 wave/SIMD"}, {1024, 256, "4 waves/SIMD"}, {2048, 256, "8 waves/SIMD"}};
    for (auto& c : cfgs) {
        double ms = time_kernel([&] { hipLaunchKernelGGL((k_addchain<P>), dim3(c.blocks), dim3(c.threads), 0, 0, d_in, d_out, iters); });
        double ms2 = time_kernel([&] { hipLaunchKernelGGL((k_maddchain<P>), dim3(c.blocks), dim3(c.threads), 0, 0, d_in, d_out, iters); });
        double n = (double)c.blocks * c.threads;
        printf("[%s] %-14s add: %8.3f ms (%6.2f us/add/chain, %7.2f M add/s)   madd: %8.3f ms (%6.2f us/madd, %7.2f M madd/s)\n", tag, c.name, ms,
               ms * 1e3 / iters, n * iters / ms / 1e3, ms2, ms2 * 1e3 / iters, n * iters / ms2 / 1e3);
    }
    return 0;
}

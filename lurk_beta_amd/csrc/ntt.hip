// ntt.hip - radix-2 NTT over the Pasta fields for gfx950.
//
// No reference counterpart: lurk-beta / arecibo are sum-check based and contain no NTT
// (SURVEY.md section 0.5); parity is against the textbook oracle only ("parity unpinned").
// omega_n = (5^((p-1)/2^32))^(2^(32-log_n)); natural order in, natural order out; the inverse
// transform uses omega^-1 and scales by 1/n.
//
// Shape: decimation-in-time after a bit-reversal gather.  Stages are fused LDS_LOG at a time: a
// workgroup loads a tile of 2^LDS_LOG elements whose indices differ only in the bits the fused
// stages touch, runs those butterflies out of LDS (twiddles from a resident table), and writes the
// tile back, so an n = 2^24 transform makes ceil(24/LDS_LOG) passes over HBM instead of 24.
#include <memory>
#include <tuple>

#include "common.hpp"
#include "field.cuh"

namespace lurk {

constexpr int NTT_LDS_LOG = 10;  // 1024 elements x 32 B = 32 KiB per workgroup
constexpr int NTT_BLOCK = 256;

template <class F>
__device__ Fe<F> ntt_pow(Fe<F> base, uint64_t e) {
    Fe<F> acc = fe_one<F>();
    while (e) {
        if (e & 1) acc = fe_mul<F>(acc, base);
        base = fe_sqr<F>(base);
        e >>= 1;
    }
    return acc;
}

// tw[i] = omega^i, i < n/2 (Montgomery)
template <class F>
__global__ __launch_bounds__(256) void ntt_twiddle_kernel(Fe<F> omega, size_t half, Fe<F>* tw) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < half) tw[i] = ntt_pow<F>(omega, i);
}

// dst[bitrev(i)] = to_mont(src[i])  (out of place)
template <class F>
__global__ __launch_bounds__(256) void ntt_bitrev_kernel(const Fe<F>* __restrict__ src, Fe<F>* __restrict__ dst, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >> log_n) return;
    size_t r = log_n ? (__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
    dst[r] = fe_to_mont<F>(src[i]);
}

// Fused stages [s0, s0+ns): stage s (1-based span m = 2^s) pairs indices differing in bit s-1.
// Tile = all indices sharing the bits outside [s0, s0+ns): local index bits map to global bits
// s0 .. s0+ns-1; the remaining global bits come from the tile id (low part below s0, high above).
template <class F>
__global__ __launch_bounds__(NTT_BLOCK) void ntt_stages_kernel(Fe<F>* __restrict__ a, const Fe<F>* __restrict__ tw, unsigned log_n, unsigned s0,
                                                                 unsigned ns, Fe<F> scale, int do_scale, int out_canonical) {
    extern __shared__ uint4 lds_raw[];
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(lds_raw);
    const size_t tile = blockIdx.x;
    const size_t low_mask = ((size_t)1 << s0) - 1;
    const size_t tile_low = tile & low_mask, tile_high = tile >> s0;
    const unsigned tile_n = 1u << ns;
    auto gidx = [&](unsigned l) -> size_t { return (tile_high << (s0 + ns)) | ((size_t)l << s0) | tile_low; };
    for (unsigned l = threadIdx.x; l < tile_n; l += NTT_BLOCK) sh[l] = a[gidx(l)];
    __syncthreads();
    for (unsigned st = 0; st < ns; st++) {
        const unsigned s = s0 + st + 1;  // global stage, span 2^s
        for (unsigned bf = threadIdx.x; bf < tile_n / 2; bf += NTT_BLOCK) {
            unsigned lo_bits = bf & ((1u << st) - 1);
            unsigned l0 = ((bf >> st) << (st + 1)) | lo_bits, l1 = l0 | (1u << st);
            // position inside the span of stage s: global index modulo 2^(s-1)
            size_t j = (((size_t)lo_bits) << s0) | tile_low;
            Fe<F> w = tw[j << (log_n - s)];
            Fe<F> u = sh[l0], t = fe_mul<F>(sh[l1], w);
            sh[l0] = fe_add<F>(u, t);
            sh[l1] = fe_sub<F>(u, t);
        }
        __syncthreads();
    }
    for (unsigned l = threadIdx.x; l < tile_n; l += NTT_BLOCK) {
        Fe<F> v = sh[l];
        if (do_scale) v = fe_mul<F>(v, scale);
        if (out_canonical) v = fe_from_mont<F>(v);
        a[gidx(l)] = v;
    }
}

template <class F>
static Fe<F> host_omega(unsigned log_n, bool inverse) {
    Fe<F> w;
    for (int i = 0; i < 8; i++) w.l[i] = F::w32(i);
    for (unsigned i = log_n; i < 32; i++) w = fe_sqr<F>(w);
    if (inverse) w = fe_inv<F>(w);
    return w;
}

// Twiddle tables and the bit-reversal scratch buffer are cached per (device, field, log_n, direction):
// generating n/2 powers costs more field multiplications than the transform itself.
struct NttPlan {
    DevBuf tw, tmp;
    std::mutex mu;
};
static std::mutex g_plan_mu;
static std::map<std::tuple<int, int, unsigned, bool>, std::unique_ptr<NttPlan>> g_plans;

template <class F>
static NttPlan& ntt_plan(unsigned log_n, bool inverse, hipStream_t s) {
    int dev = 0;
    LURK_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto key = std::make_tuple(dev, (int)F::ID, log_n, inverse);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        const size_t n = (size_t)1 << log_n;
        auto plan = std::make_unique<NttPlan>();
        plan->tw.alloc((n / 2 ? n / 2 : 1) * 32);
        plan->tmp.alloc(n * 32);
        if (n > 1) {
            Fe<F> omega = host_omega<F>(log_n, inverse);
            hipLaunchKernelGGL((ntt_twiddle_kernel<F>), dim3(div_up(n / 2, 256)), dim3(256), 0, s, omega, n / 2, plan->tw.template as<Fe<F>>());
            LURK_HIP_CHECK(hipGetLastError());
            LURK_HIP_CHECK(hipStreamSynchronize(s));
        }
        it = g_plans.emplace(key, std::move(plan)).first;
    }
    return *it->second;
}

template <class F>
static void ntt_device(void* d_data, unsigned log_n, bool inverse, hipStream_t s) {
    LURK_REQUIRE(log_n <= 28, "log_n too large");
    const size_t n = (size_t)1 << log_n;
    NttPlan& plan = ntt_plan<F>(log_n, inverse, s);
    std::lock_guard<std::mutex> lk(plan.mu);  // one transform at a time per plan (shared scratch)
    Fe<F>* tmp = plan.tmp.template as<Fe<F>>();
    const Fe<F>* tw = plan.tw.template as<Fe<F>>();
    ProfScope ps("ntt", s);
    hipLaunchKernelGGL((ntt_bitrev_kernel<F>), dim3(div_up(n, 256)), dim3(256), 0, s, (const Fe<F>*)d_data, tmp, log_n);
    Fe<F> scale = fe_one<F>();
    if (inverse) scale = fe_inv<F>(fe_from_u64<F>((uint64_t)n));
    unsigned s0 = 0;
    if (log_n == 0) hipLaunchKernelGGL((ntt_stages_kernel<F>), dim3(1), dim3(NTT_BLOCK), 32, s, tmp, tw, log_n, 0u, 0u, scale, 0, 1);
    while (s0 < log_n) {
        unsigned ns = log_n - s0 < (unsigned)NTT_LDS_LOG ? log_n - s0 : NTT_LDS_LOG;
        bool last = s0 + ns == log_n;
        hipLaunchKernelGGL((ntt_stages_kernel<F>), dim3((unsigned)(n >> ns)), dim3(NTT_BLOCK), ((size_t)32 << ns), s, tmp, tw, log_n, s0, ns,
                           scale, (last && inverse) ? 1 : 0, last ? 1 : 0);
        s0 += ns;
    }
    LURK_HIP_CHECK(hipGetLastError());
    LURK_HIP_CHECK(hipMemcpyAsync(d_data, tmp, n * 32, hipMemcpyDeviceToDevice, s));
    LURK_HIP_CHECK(hipStreamSynchronize(s));  // the plan's scratch is released to the next caller
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_ntt_dev(int field_id, void* d_inout, unsigned log_n, int inverse, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id == 0 || field_id == 1, "NTT is offered over the Pasta fields only");
        LURK_REQUIRE(d_inout, "null buffer");
        if (field_id == 0) ntt_device<PallasFp>(d_inout, log_n, inverse != 0, (hipStream_t)stream);
        else ntt_device<PallasFq>(d_inout, log_n, inverse != 0, (hipStream_t)stream);
    });
}
int lurk_hip_ntt(int field_id, void* inout, unsigned log_n, int inverse) {
    return guarded([&] {
        LURK_REQUIRE(field_id == 0 || field_id == 1, "NTT is offered over the Pasta fields only");
        LURK_REQUIRE(inout, "null buffer");
        LURK_REQUIRE(log_n <= 28, "log_n too large");
        size_t n = (size_t)1 << log_n;
        DevBuf d(n * 32);
        LURK_HIP_CHECK(hipMemcpy(d.p, inout, n * 32, hipMemcpyHostToDevice));
        if (field_id == 0) ntt_device<PallasFp>(d.p, log_n, inverse != 0, nullptr);
        else ntt_device<PallasFq>(d.p, log_n, inverse != 0, nullptr);
        LURK_HIP_CHECK(hipMemcpy(inout, d.p, n * 32, hipMemcpyDeviceToHost));
    });
}
}

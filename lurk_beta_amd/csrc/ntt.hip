// ntt.hip - radix-2 NTT over the Pasta fields for gfx950.
//
// No reference counterpart: lurk-beta / arecibo are sum-check based and contain no NTT
// (SURVEY.md section 0.5); parity is against the textbook oracle only ("parity unpinned").
// omega_n = (5^((p-1)/2^32))^(2^(32-log_n)); natural order in, natural order out; the inverse
// transform uses omega^-1 and scales by 1/n.
//
// Shape: decimation-in-time after a bit-reversal permutation, in passes that each fuse several
// radix-2 stages out of an LDS tile:
//   * the permutation is a 32x32 LDS transpose (both the reads and the writes are 1 KiB rows);
//   * pass 1 (stages 1..11) works on 2048 contiguous elements;
//   * a later pass covering stages s0+1..s0+ns takes, per workgroup, C = 2048 / 2^ns neighbouring
//     sub-transforms so that every global row is C x 32 B contiguous, "twists" each element once by
//     omega_N^(t * rev(l)) (N = 2^(s0+ns), t = position inside the already transformed 2^s0 block),
//     after which every stage only needs the 2^(ns-1) local twiddles, kept in LDS;
//   * the last pass writes straight into the caller's buffer (scaled / converted as needed).
#include <memory>
#include <tuple>

#include "common.hpp"
#include "field.cuh"

namespace lurk {


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

// dst[bitrev(i)] = to_mont(src[i])  (out of place), small sizes
template <class F>
__global__ __launch_bounds__(256) void ntt_bitrev_kernel(const Fe<F>* __restrict__ src, Fe<F>* __restrict__ dst, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >> log_n) return;
    size_t r = log_n ? (__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
    dst[r] = fe_to_mont<F>(src[i]);
}
// same for log_n >= 10 as a tiled transpose: index bits [hi:5][mid][lo:5] -> [rev lo][rev mid][rev hi];
// block = one value of mid; reads rows of 32 consecutive lo, writes rows of 32 consecutive rev(hi)
template <class F>
__global__ __launch_bounds__(256) void ntt_bitrev_tiled_kernel(const Fe<F>* __restrict__ src, Fe<F>* __restrict__ dst, unsigned log_n) {
    __shared__ uint4 lds_raw[32 * 33 * 2];
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(lds_raw);
    const unsigned mbits = log_n - 10;
    const size_t mid = blockIdx.x;
    const size_t rmid = mbits ? (__brevll((unsigned long long)mid) >> (64 - mbits)) : 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned e = r * 256 + threadIdx.x, hi = e >> 5, lo = e & 31;
        sh[lo * 33 + hi] = fe_to_mont<F>(src[((size_t)hi << (log_n - 5)) | (mid << 5) | lo]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned e = r * 256 + threadIdx.x, a = e >> 5, b = e & 31;  // a = rev5(lo), b = rev5(hi)
        unsigned lo = __brev(a) >> 27, hi = __brev(b) >> 27;
        dst[((size_t)a << (log_n - 5)) | (rmid << 5) | b] = sh[lo * 33 + hi];
    }
}

constexpr int NTT_TILE_LOG = 11;      // 2048 elements (64 KiB) per workgroup
constexpr int NTT_PASS_BLOCK = 512;

// One pass: stages s0+1 .. s0+ns.  cbits = log2(C), ns + cbits <= NTT_TILE_LOG.  Tile layout in LDS:
// [l][c] (l = index inside the sub-transform, c = which of the C neighbouring sub-transforms).
template <class F>
__global__ __launch_bounds__(NTT_PASS_BLOCK) void ntt_pass_kernel(const Fe<F>* __restrict__ in, Fe<F>* __restrict__ out,
                                                                    const Fe<F>* __restrict__ tw, unsigned log_n, unsigned s0, unsigned ns,
                                                                    unsigned cbits, Fe<F> scale, int do_scale, int out_canonical) {
    extern __shared__ uint4 lds_raw[];
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(lds_raw);
    Fe<F>* lt = sh + ((size_t)1 << (ns + cbits));  // local twiddles: omega_{2^ns}^j, j < 2^(ns-1)
    const unsigned C = 1u << cbits, tile_n = 1u << (ns + cbits);
    const size_t half_n = (size_t)1 << (log_n - 1);
    // block id -> (H, tbase): the 2^s0 low positions are cut into groups of C
    const size_t groups = ((size_t)1 << s0) >> cbits;  // >= 1
    const size_t H = blockIdx.x / groups, tbase = (blockIdx.x % groups) << cbits;
    for (unsigned j = threadIdx.x; j < (1u << ns) / 2; j += NTT_PASS_BLOCK) lt[j] = tw[(size_t)j << (log_n - ns)];
    for (unsigned e = threadIdx.x; e < tile_n; e += NTT_PASS_BLOCK) {
        unsigned c = e & (C - 1), l = e >> cbits;
        size_t t = tbase + c;
        Fe<F> v = in[(H << (s0 + ns)) | ((size_t)l << s0) | t];
        if (s0) {  // twist by omega_N^(t * rev_ns(l)), N = 2^(s0+ns): index into the omega^i table with the sign fold
            size_t idx = (t * (size_t)(__brev(l) >> (32 - ns))) << (log_n - s0 - ns);
            Fe<F> w = tw[idx & (half_n - 1)];
            if (idx & half_n) w = fe_neg<F>(w);
            v = fe_mul<F>(v, w);
        }
        sh[e] = v;  // e == l * C + c
    }
    __syncthreads();
    for (unsigned st = 0; st < ns; st++) {
        for (unsigned bf = threadIdx.x; bf < tile_n / 2; bf += NTT_PASS_BLOCK) {
            unsigned c = bf & (C - 1), pr = bf >> cbits;
            unsigned lo_bits = pr & ((1u << st) - 1);
            unsigned l0 = ((pr >> st) << (st + 1)) | lo_bits, l1 = l0 | (1u << st);
            Fe<F> w = lt[lo_bits << (ns - st - 1)];
            Fe<F> u = sh[(l0 << cbits) | c], x = fe_mul<F>(sh[(l1 << cbits) | c], w);
            sh[(l0 << cbits) | c] = fe_add<F>(u, x);
            sh[(l1 << cbits) | c] = fe_sub<F>(u, x);
        }
        __syncthreads();
    }
    for (unsigned e = threadIdx.x; e < tile_n; e += NTT_PASS_BLOCK) {
        unsigned c = e & (C - 1), l = e >> cbits;
        Fe<F> v = sh[e];
        if (do_scale) v = fe_mul<F>(v, scale);
        if (out_canonical) v = fe_from_mont<F>(v);
        out[(H << (s0 + ns)) | ((size_t)l << s0) | (tbase + c)] = v;
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
    DevBuf tw;
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
    // the ping-pong scratch is stream-ordered (allocated and freed on `s`): concurrent transforms on one plan never
    // share it and the call does not synchronise
    Fe<F>* tmp = nullptr;
    LURK_HIP_CHECK(hipMallocAsync((void**)&tmp, n * 32, s));
    Fe<F>* data = (Fe<F>*)d_data;
    const Fe<F>* tw = plan.tw.template as<Fe<F>>();
    allow_dynamic_lds((const void*)ntt_pass_kernel<F>, 160 * 1024);
    ProfScope ps("ntt", s);
    if (log_n >= 10) hipLaunchKernelGGL((ntt_bitrev_tiled_kernel<F>), dim3((unsigned)(n >> 10)), dim3(256), 0, s, data, tmp, log_n);
    else hipLaunchKernelGGL((ntt_bitrev_kernel<F>), dim3(div_up(n, 256)), dim3(256), 0, s, data, tmp, log_n);
    Fe<F> scale = fe_one<F>();
    if (inverse) scale = fe_inv<F>(fe_from_u64<F>((uint64_t)n));
    // pass plan: first pass as deep as the tile allows, the rest in passes of <= 9 stages (C >= 4: rows of >= 128 B)
    unsigned s0 = 0;
    bool first = true;
    do {
        unsigned left = log_n - s0;
        unsigned ns = first ? (left < (unsigned)NTT_TILE_LOG ? left : NTT_TILE_LOG) : (left <= 9 ? left : ((left + 1) / 2 < 8 ? (left + 1) / 2 : 8));
        unsigned cbits = first ? 0 : (NTT_TILE_LOG - ns < s0 ? NTT_TILE_LOG - ns : s0);
        bool last = s0 + ns == log_n;
        size_t lds = (((size_t)1 << (ns + cbits)) + ((size_t)1 << ns) / 2 + 1) * sizeof(Fe<F>);
        hipLaunchKernelGGL((ntt_pass_kernel<F>), dim3((unsigned)(n >> (ns + cbits))), dim3(NTT_PASS_BLOCK), lds, s, tmp, last ? data : tmp, tw,
                           log_n, s0, ns, cbits, scale, (last && inverse) ? 1 : 0, last ? 1 : 0);
        s0 += ns;
        first = false;
    } while (s0 < log_n);
    hipError_t launch_err = hipGetLastError();
    (void)hipFreeAsync(tmp, s);
    LURK_HIP_CHECK(launch_err);
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
        LURK_HIP_CHECK(hipStreamSynchronize(nullptr));
        LURK_HIP_CHECK(hipMemcpy(inout, d.p, n * 32, hipMemcpyDeviceToHost));
    });
}
}

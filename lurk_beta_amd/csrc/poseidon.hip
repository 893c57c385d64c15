// poseidon.hip - Poseidon batch hasher and dense arity-8 tree builder for gfx950.
//
// Kernel shape (north_star): one Poseidon state per lane (t x 9 VGPRs: radix-2^29 limbs, poseidon29.cuh),
// the whole constant image (round constants, MDS, pre-sparse and sparse matrices; 59 KiB for t = 9) staged once per
// workgroup in LDS and read back as wave-uniform broadcasts (every lane of a wave is in the same
// round, so every ds_read hits one address: conflict-free).  No MFMA: the work is 255-bit modular
// multiplication on the integer VALU (v_mad_u64_u32), ~2.1k field multiplications per hash8.
//
// Replaces: PoseidonCache::hash3/4/6/8 -> neptune Poseidon::hash (/root/reference/src/hash.rs:180-204)
// and the hash8 tree of coprocessor::trie (/root/reference/src/coprocessor/trie/mod.rs:434-481).
#include <memory>

#include "common.hpp"
#include "poseidon.cuh"
#include "poseidon29.cuh"
#include "poseidon_params.hpp"

namespace lurk {

constexpr int POSEIDON_BLOCK = 256;
constexpr size_t POSEIDON_WIDE_MAX = 8192;  // batches up to this size run one state across T lanes (latency), larger ones one per lane
enum : int { PF_IN_MONT = 1, PF_OUT_MONT = 2 };

template <class P, int T>
__global__ __launch_bounds__(POSEIDON_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void poseidon_batch_kernel(const uint4* __restrict__ pre, uint4* __restrict__ out,
                                                                          size_t n, const uint4* __restrict__ img, int img_vec4,
                                                                          int rf, int rp, int flags) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < img_vec4; i += POSEIDON_BLOCK) lds[i] = img[i];
    __syncthreads();
    const uint32_t* C = reinterpret_cast<const uint32_t*>(lds);
    const uint32_t* mont2 = C + (size_t)PoseidonLayout<T>(rf, rp).total() * P29_STRIDE;  // 2^522 mod p
    constexpr int A = T - 1;
    for (size_t h = (size_t)blockIdx.x * POSEIDON_BLOCK + threadIdx.x; h < n; h += (size_t)gridDim.x * POSEIDON_BLOCK) {
        F29<P> s[T];
        s[0] = ld_const29<P>(C);
        const uint4* src = pre + h * (A * 2);
#pragma unroll
        for (int i = 0; i < A; i++) {
            uint4 lo = src[2 * i], hi = src[2 * i + 1];
            Fe<P> x;
            x.l[0] = lo.x; x.l[1] = lo.y; x.l[2] = lo.z; x.l[3] = lo.w;
            x.l[4] = hi.x; x.l[5] = hi.y; x.l[6] = hi.z; x.l[7] = hi.w;
            s[i + 1] = (flags & PF_IN_MONT) ? f29_from_mont256<P>(x) : poseidon29_from_canonical<P>(x.l, mont2);
        }
        poseidon29_permute<P, T>(s, C, rf, rp);
        Fe<P> d = (flags & PF_OUT_MONT) ? f29_to_mont256<P>(s[1]) : poseidon29_to_canonical<P>(s[1]);
        out[2 * h] = make_uint4(d.l[0], d.l[1], d.l[2], d.l[3]);
        out[2 * h + 1] = make_uint4(d.l[4], d.l[5], d.l[6], d.l[7]);
    }
}

// ---- small batches: one state across T lanes -----------------------------------------------------------------
// A single lane needs ~1 600 dependent products per hash8 (0.57 ms), so a batch that does not fill the chip
// (the top levels of a tree, a trie path, a store-hydration level) is latency-bound.  Here lane e of a group of
// T lanes owns state element e (64 / T hashes per wave):
//   full round     S-boxes in parallel; row e of the matrix is accumulated by lane e from the group's elements
//                  (fetched lane to lane), one reduction
//   partial round  x = sbox(s_0) is broadcast; lane e forms its term of the sparse row (x n00 | s_e v_e), the T
//                  reduced terms are summed across the group into lane 0; lane e >= 1 adds x w_e to its element
// ~450 product-times per hash instead of ~1 600.  Same constants, same image, bit-identical digests.
template <class P>
__device__ __forceinline__ F29<P> grp_fetch(const F29<P>& v, int src_lane) {
    F29<P> r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = __shfl(v.l[k], src_lane);
    return r;
}

constexpr int POSEIDON_WIDE_BLOCK = 256;

template <class P, int T>
__global__ __launch_bounds__(POSEIDON_WIDE_BLOCK) void poseidon_wide_kernel(const uint4* __restrict__ pre, uint4* __restrict__ out, size_t n,
                                                                              const uint4* __restrict__ img, int img_vec4, int rf, int rp,
                                                                              int flags) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < img_vec4; i += POSEIDON_WIDE_BLOCK) lds[i] = img[i];
    __syncthreads();
    const uint32_t* C = reinterpret_cast<const uint32_t*>(lds);
    const PoseidonLayout<T> L(rf, rp);
    const uint32_t* mont2 = C + (size_t)L.total() * P29_STRIDE;
    constexpr int G = 64 / T, A = T - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_group = lane < G * T;
    const int g = in_group ? lane / T : 0, e = in_group ? lane % T : 0;  // spare lanes shadow lane 0 and never store
    const int base = g * T;
    const size_t h = ((size_t)blockIdx.x * (POSEIDON_WIDE_BLOCK / 64) + wave) * G + g;
    const bool live = in_group && h < n;

    F29<P> s;
    if (e == 0) {
        s = ld_const29<P>(C);
    } else {
        Fe<P> x = fe_zero<P>();
        if (live) {
            const uint4* src = pre + (h * A + (e - 1)) * 2;
            uint4 lo = src[0], hi = src[1];
            x.l[0] = lo.x; x.l[1] = lo.y; x.l[2] = lo.z; x.l[3] = lo.w;
            x.l[4] = hi.x; x.l[5] = hi.y; x.l[6] = hi.z; x.l[7] = hi.w;
        }
        s = (flags & PF_IN_MONT) ? f29_from_mont256<P>(x) : poseidon29_from_canonical<P>(x.l, mont2);
    }

    auto full_round = [&](const uint32_t* rc, const uint32_t* mat) {
        s = f29_pow5<P>(f29_add<P>(s, ld_const29<P>(rc + e * P29_STRIDE)));
        Dot29<P> acc;
        dot29_init<P>(acc);
#pragma unroll
        for (int i = 0; i < T; i++) {
            if (T > 5 && i == 4) dot29_carry<P>(acc);
            dot29_mac<P>(acc, grp_fetch<P>(s, base + i), ld_const29<P>(mat + (e * T + i) * P29_STRIDE));
        }
        s = dot29_finish<P>(acc);
    };

    const uint32_t* mds = C + L.mds() * P29_STRIDE;
#pragma unroll 1
    for (int r = 0; r < L.h; r++) full_round(C + (L.rc1() + r * T) * P29_STRIDE, r == L.h - 1 ? C + L.pre() * P29_STRIDE : mds);
#pragma unroll 1
    for (int p = 0; p < rp; p++) {
        const uint32_t* sp = C + (L.sp() + p * (2 * T - 1)) * P29_STRIDE;
        const F29<P> xl = f29_pow5<P>(f29_add<P>(s, ld_const29<P>(C + (L.pk() + p) * P29_STRIDE)));  // meaningful on e == 0
        const F29<P> x = grp_fetch<P>(xl, base);
        F29<P> a;
#pragma unroll
        for (int k = 0; k < 9; k++) a.l[k] = e == 0 ? x.l[k] : s.l[k];
        F29<P> term = f29_mul<P>(a, ld_const29<P>(sp + e * P29_STRIDE));  // tight, < 2^254.2
#pragma unroll
        for (int off = 1; off < T; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                uint32_t t = __shfl_down(term.l[k], off);
                if (e + off < T) term.l[k] += t;
            }
            if (off == 2 || off * 2 >= T) term = f29_carry<P>(term);  // at most 4 tight values between carries
        }
        const F29<P> upd = f29_carry<P>(f29_add<P>(s, f29_mul<P>(x, ld_const29<P>(sp + (T - 1 + e) * P29_STRIDE))));
#pragma unroll
        for (int k = 0; k < 9; k++) s.l[k] = e == 0 ? term.l[k] : upd.l[k];
    }
#pragma unroll 1
    for (int r = 0; r < L.h; r++) full_round(r == 0 ? C + L.after() * P29_STRIDE : C + (L.rc2() + (r - 1) * T) * P29_STRIDE, mds);

    if (live && e == 1) {
        Fe<P> d = (flags & PF_OUT_MONT) ? f29_to_mont256<P>(s) : poseidon29_to_canonical<P>(s);
        out[2 * h] = make_uint4(d.l[0], d.l[1], d.l[2], d.l[3]);
        out[2 * h + 1] = make_uint4(d.l[4], d.l[5], d.l[6], d.l[7]);
    }
}

// ---- per-(field, arity) constants, generated on the host once and kept in HBM ---------------
struct PoseidonConsts {
    int rf = 0, rp = 0, t = 0;
    std::vector<uint32_t> rc_canon, mds_canon;  // for lurk_hip_poseidon_constants
    std::vector<uint32_t> image;
    std::map<int, DevBuf> dev;  // device id -> image
};
static std::mutex g_pc_mu;
static std::map<std::pair<int, int>, std::unique_ptr<PoseidonConsts>> g_pc;

template <class P>
static std::unique_ptr<PoseidonConsts> build_consts(int arity) {
    PoseidonParams<P> pp = make_poseidon_params<P>(arity);
    auto pc = std::make_unique<PoseidonConsts>();
    pc->rf = pp.rf;
    pc->rp = pp.rp;
    pc->t = pp.t;
    pc->image = poseidon29_image<P>(poseidon_device_image<P>(pp));  // radix-2^29 form, 12 words per constant
    for (auto& x : pp.rc) {
        Fe<P> c = fe_from_mont<P>(x);
        for (int i = 0; i < 8; i++) pc->rc_canon.push_back(c.l[i]);
    }
    for (auto& x : pp.mds) {
        Fe<P> c = fe_from_mont<P>(x);
        for (int i = 0; i < 8; i++) pc->mds_canon.push_back(c.l[i]);
    }
    return pc;
}

static PoseidonConsts& get_consts(int field_id, int arity) {
    LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
    LURK_REQUIRE(arity == 3 || arity == 4 || arity == 6 || arity == 8, "unsupported arity (must be 3, 4, 6 or 8)");
    std::lock_guard<std::mutex> lk(g_pc_mu);
    auto key = std::make_pair(field_id, arity);
    auto it = g_pc.find(key);
    if (it == g_pc.end()) {
        std::unique_ptr<PoseidonConsts> pc;
        if (field_id == 0) pc = build_consts<PallasFp>(arity);
        else if (field_id == 1) pc = build_consts<PallasFq>(arity);
        else pc = build_consts<Bn254Fr>(arity);
        it = g_pc.emplace(key, std::move(pc)).first;
    }
    return *it->second;
}

static const uint4* device_image(PoseidonConsts& pc, hipStream_t s) {
    int dev = 0;
    LURK_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_pc_mu);
    auto it = pc.dev.find(dev);
    if (it == pc.dev.end()) {
        DevBuf b(pc.image.size() * 4);
        LURK_HIP_CHECK(hipMemcpy(b.p, pc.image.data(), pc.image.size() * 4, hipMemcpyHostToDevice));
        it = pc.dev.emplace(dev, std::move(b)).first;
    }
    return it->second.as<uint4>();
}

template <class P, int T>
static void launch_batch(const void* d_pre, void* d_out, size_t n, PoseidonConsts& pc, int flags, hipStream_t s) {
    if (n == 0) return;
    const uint4* img = device_image(pc, s);
    int img_vec4 = (int)(pc.image.size() / 4);
    size_t lds_bytes = (size_t)img_vec4 * 16;
    auto kern = poseidon_batch_kernel<P, T>;
    if (n <= POSEIDON_WIDE_MAX) {  // does not fill the chip: trade throughput for latency
        auto wide = poseidon_wide_kernel<P, T>;
        allow_dynamic_lds((const void*)wide, (int)lds_bytes);
        constexpr size_t per_block = (size_t)(POSEIDON_WIDE_BLOCK / 64) * (64 / T);
        ProfScope ps("poseidon_batch", s);
        hipLaunchKernelGGL(wide, dim3(div_up(n, per_block)), dim3(POSEIDON_WIDE_BLOCK), lds_bytes, s, (const uint4*)d_pre, (uint4*)d_out, n, img,
                           img_vec4, pc.rf, pc.rp, flags);
        LURK_HIP_CHECK(hipGetLastError());
        return;
    }
    allow_dynamic_lds((const void*)kern, (int)lds_bytes);
    unsigned blocks = div_up(n, POSEIDON_BLOCK);
    unsigned cap = (unsigned)num_cus() * 2;  // 2 workgroups (8 waves) per CU, grid-stride beyond
    if (blocks > cap) blocks = cap;
    ProfScope ps("poseidon_batch", s);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(POSEIDON_BLOCK), lds_bytes, s, (const uint4*)d_pre, (uint4*)d_out, n, img, img_vec4,
                       pc.rf, pc.rp, flags);
    LURK_HIP_CHECK(hipGetLastError());
}

template <class P>
static void launch_batch_f(int arity, const void* d_pre, void* d_out, size_t n, PoseidonConsts& pc, int flags, hipStream_t s) {
    switch (arity) {
        case 3: launch_batch<P, 4>(d_pre, d_out, n, pc, flags, s); break;
        case 4: launch_batch<P, 5>(d_pre, d_out, n, pc, flags, s); break;
        case 6: launch_batch<P, 7>(d_pre, d_out, n, pc, flags, s); break;
        case 8: launch_batch<P, 9>(d_pre, d_out, n, pc, flags, s); break;
    }
}

void poseidon_batch_device(int field_id, int arity, const void* d_pre, void* d_out, size_t n, int flags, hipStream_t s) {
    PoseidonConsts& pc = get_consts(field_id, arity);
    if (field_id == 0) launch_batch_f<PallasFp>(arity, d_pre, d_out, n, pc, flags, s);
    else if (field_id == 1) launch_batch_f<PallasFq>(arity, d_pre, d_out, n, pc, flags, s);
    else launch_batch_f<Bn254Fr>(arity, d_pre, d_out, n, pc, flags, s);
}

static bool is_pow8(size_t n) {
    if (n < 8) return false;
    while (n % 8 == 0) n /= 8;
    return n == 1;
}

// Level-synchronous tree: every level is one batch launch on the same stream; levels stay in HBM
// (inner levels are kept in Montgomery form between launches and converted in place at the end
// only if the caller asked for them... simpler and exact: each level is written canonical).
void poseidon_tree8_device(int field_id, const void* d_leaves, size_t n_leaves, void* d_levels, hipStream_t s) {
    LURK_REQUIRE(is_pow8(n_leaves), "n_leaves must be a power of 8 (>= 8)");
    const char* in = (const char*)d_leaves;
    char* out = (char*)d_levels;
    size_t cur = n_leaves;
    while (cur > 1) {
        size_t next = cur / 8;
        poseidon_batch_device(field_id, 8, in, out, next, 0, s);
        in = out;
        out += next * 32;
        cur = next;
    }
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_poseidon_constants(int field_id, int arity, int* rf, int* rp, void* rc, void* mds) {
    // pure host computation: usable without a device (tests compare it with the oracle on CPU)
    try {
        PoseidonConsts& pc = get_consts(field_id, arity);
        if (rf) *rf = pc.rf;
        if (rp) *rp = pc.rp;
        if (rc) memcpy(rc, pc.rc_canon.data(), pc.rc_canon.size() * 4);
        if (mds) memcpy(mds, pc.mds_canon.data(), pc.mds_canon.size() * 4);
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}

int lurk_hip_poseidon_batch_dev(int field_id, int arity, const void* d_preimages, size_t n, void* d_digests, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(n == 0 || (d_preimages && d_digests), "null buffer");
        poseidon_batch_device(field_id, arity, d_preimages, d_digests, n, 0, (hipStream_t)stream);
    });
}

int lurk_hip_poseidon_batch(int field_id, int arity, const void* preimages, size_t n, void* digests) {
    return guarded([&] {
        get_consts(field_id, arity);  // validates arguments first
        if (n == 0) return;
        LURK_REQUIRE(preimages && digests, "null buffer");
        DevBuf in(n * arity * 32), out(n * 32);
        LURK_HIP_CHECK(hipMemcpy(in.p, preimages, n * arity * 32, hipMemcpyHostToDevice));
        poseidon_batch_device(field_id, arity, in.p, out.p, n, 0, nullptr);
        LURK_HIP_CHECK(hipMemcpy(digests, out.p, n * 32, hipMemcpyDeviceToHost));
    });
}

int lurk_hip_poseidon_tree8_dev(int field_id, const void* d_leaves, size_t n_leaves, void* d_levels, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(d_leaves && d_levels, "null buffer");
        poseidon_tree8_device(field_id, d_leaves, n_leaves, d_levels, (hipStream_t)stream);
    });
}

int lurk_hip_poseidon_tree8(int field_id, const void* leaves, size_t n_leaves, void* root32, void* levels_or_null) {
    return guarded([&] {
        LURK_REQUIRE(leaves && root32, "null buffer");
        LURK_REQUIRE(is_pow8(n_leaves), "n_leaves must be a power of 8 (>= 8)");
        size_t internal = (n_leaves - 1) / 7;
        DevBuf in(n_leaves * 32), lv(internal * 32);
        LURK_HIP_CHECK(hipMemcpy(in.p, leaves, n_leaves * 32, hipMemcpyHostToDevice));
        poseidon_tree8_device(field_id, in.p, n_leaves, lv.p, nullptr);
        LURK_HIP_CHECK(hipMemcpy(root32, (char*)lv.p + (internal - 1) * 32, 32, hipMemcpyDeviceToHost));
        if (levels_or_null) LURK_HIP_CHECK(hipMemcpy(levels_or_null, lv.p, internal * 32, hipMemcpyDeviceToHost));
    });
}
}

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


// ---- slot witnesses -------------------------------------------------------------------
// What generate_slots_witnesses (/root/reference/src/lem/multiframe.rs:520-592) computes on the CPU, slot by slot, with
// neptune's circuit2::poseidon_hash_allocated on a WitnessCS (/root/reference/src/lem/circuit.rs:212-240): the block
//   [preimage (arity) | l^2, l^4, l^5 + key for every S-box in circuit order | digest]
// of Montgomery values that the slot contributes to the witness vector W.  Written straight into the device-resident W
// (block of slot i at element d_offsets[i], or first + i * stride), so the ~85 % of W that is Poseidon trace never
// crosses PCIe.  Same two kernel shapes as the batch hasher: T lanes per hash for the small batches of a folding step
// (every lane emits the triple of its own S-box), one hash per lane for large ones.
template <class P>
__device__ __forceinline__ void trace_store(Fe<P>* __restrict__ w, size_t idx, const F29<P>& v) {
    const Fe<P> d = f29_to_mont256<P>(v);
    uint4* dst = reinterpret_cast<uint4*>(w + idx);
    dst[0] = make_uint4(d.l[0], d.l[1], d.l[2], d.l[3]);
    dst[1] = make_uint4(d.l[4], d.l[5], d.l[6], d.l[7]);
}
struct TraceDst {
    void* w;                  // Fe<P>*: the witness vector
    const uint64_t* offsets;  // per-slot element offsets, or null
    size_t first, stride;     // ... then first + (i / group) * group_stride + (i % group) * stride
    size_t group = ~(size_t)0, group_stride = 0;  // slots per frame and frame length (frame-major slot numbering)
    __device__ __forceinline__ size_t base(size_t i) const {
        return offsets ? (size_t)offsets[i] : first + (i / group) * group_stride + (i % group) * stride;
    }
};

template <class P, int T>
__global__ __launch_bounds__(POSEIDON_WIDE_BLOCK) void poseidon_trace_wide_kernel(const uint4* __restrict__ pre, size_t n, const uint4* __restrict__ img,
                                                                                    int img_vec4, int rf, int rp, int flags, TraceDst dst) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < img_vec4; i += POSEIDON_WIDE_BLOCK) lds[i] = img[i];
    __syncthreads();
    const uint32_t* C = reinterpret_cast<const uint32_t*>(lds);
    const PoseidonLayout<T> L(rf, rp);
    const uint32_t* mont2 = C + (size_t)L.total() * P29_STRIDE;
    const uint32_t* post = mont2 + P29_STRIDE;
    constexpr int G = 64 / T, A = T - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_group = lane < G * T;
    const int g = in_group ? lane / T : 0, e = in_group ? lane % T : 0;
    const int base = g * T;
    const size_t h = ((size_t)blockIdx.x * (POSEIDON_WIDE_BLOCK / 64) + wave) * G + g;
    const bool live = in_group && h < n;
    Fe<P>* w = (Fe<P>*)dst.w;
    const size_t wb = live ? dst.base(h) : 0;

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
        if (live) trace_store<P>(w, wb + (e - 1), s);  // the preimage opens the block
    }
    int sbox = 0;  // S-boxes before the current round
    auto sbox_emit = [&](const F29<P>& l, int idx, bool mine) {
        const F29<P> l2 = f29_sqr<P>(l), l4 = f29_sqr<P>(l2), l5 = f29_mul<P>(l4, l);
        if (mine) {
            const size_t o = wb + A + 3 * (size_t)idx;
            trace_store<P>(w, o, l2);
            trace_store<P>(w, o + 1, l4);
            trace_store<P>(w, o + 2, f29_add<P>(l5, ld_const29<P>(post + (size_t)idx * P29_STRIDE)));
        }
        return l5;
    };
    auto full_round = [&](const uint32_t* rc, const uint32_t* mat) {
        s = sbox_emit(f29_carry<P>(f29_add<P>(s, ld_const29<P>(rc + e * P29_STRIDE))), sbox + e, live);
        sbox += T;
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
        const F29<P> xl = sbox_emit(f29_carry<P>(f29_add<P>(s, ld_const29<P>(C + (L.pk() + p) * P29_STRIDE))), sbox, live && e == 0);
        sbox += 1;
        const F29<P> x = grp_fetch<P>(xl, base);
        F29<P> a;
#pragma unroll
        for (int k = 0; k < 9; k++) a.l[k] = e == 0 ? x.l[k] : s.l[k];
        F29<P> term = f29_mul<P>(a, ld_const29<P>(sp + e * P29_STRIDE));
#pragma unroll
        for (int off = 1; off < T; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                uint32_t t = __shfl_down(term.l[k], off);
                if (e + off < T) term.l[k] += t;
            }
            if (off == 2 || off * 2 >= T) term = f29_carry<P>(term);
        }
        const F29<P> upd = f29_carry<P>(f29_add<P>(s, f29_mul<P>(x, ld_const29<P>(sp + (T - 1 + e) * P29_STRIDE))));
#pragma unroll
        for (int k = 0; k < 9; k++) s.l[k] = e == 0 ? term.l[k] : upd.l[k];
    }
#pragma unroll 1
    for (int r = 0; r < L.h; r++) full_round(r == 0 ? C + L.after() * P29_STRIDE : C + (L.rc2() + (r - 1) * T) * P29_STRIDE, mds);
    if (live && e == 1) trace_store<P>(w, wb + A + 3 * (size_t)sbox, s);  // the digest closes the block
}

template <class P, int T>
__global__ __launch_bounds__(POSEIDON_BLOCK) void poseidon_trace_batch_kernel(const uint4* __restrict__ pre, size_t n, const uint4* __restrict__ img,
                                                                                int img_vec4, int rf, int rp, int flags, TraceDst dst) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < img_vec4; i += POSEIDON_BLOCK) lds[i] = img[i];
    __syncthreads();
    const uint32_t* C = reinterpret_cast<const uint32_t*>(lds);
    const PoseidonLayout<T> L(rf, rp);
    const uint32_t* mont2 = C + (size_t)L.total() * P29_STRIDE;
    const uint32_t* post = mont2 + P29_STRIDE;
    constexpr int A = T - 1;
    Fe<P>* w = (Fe<P>*)dst.w;
    for (size_t h = (size_t)blockIdx.x * POSEIDON_BLOCK + threadIdx.x; h < n; h += (size_t)gridDim.x * POSEIDON_BLOCK) {
        const size_t wb = dst.base(h);
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
            trace_store<P>(w, wb + i, s[i + 1]);
        }
        auto emit = [&](int sbox, const F29<P>& l2, const F29<P>& l4, const F29<P>& l5k) {
            const size_t o = wb + A + 3 * (size_t)sbox;
            trace_store<P>(w, o, l2);
            trace_store<P>(w, o + 1, l4);
            trace_store<P>(w, o + 2, l5k);
        };
        poseidon29_permute_trace<P, T>(s, C, post, rf, rp, emit);
        trace_store<P>(w, wb + A + 3 * ((size_t)T * rf + rp), s[1]);
    }
}

// Bit-decomposition slots: bellpepper's AllocatedNum::to_bits_le_strict as allocate_img_for_slot runs it
// (/root/reference/src/lem/circuit.rs:236-238).  The allocation order depends only on the field: walking the bits of p - 1
// from the top, a 1-bit allocates the value's bit, a 0-bit first closes the open run of 1-bits with a chain of ANDs (one
// aux each, chained with the previous run's result) and then allocates the value's bit.  Every aux is therefore either
// "bit i of v" or "v has all the bits of mask m": the host compiles the field's walk into that program once
// (slot sizes 298 / 301 / 354 = the reference's own constants, multiframe.rs:495-497).
struct BitDecompProgram {
    std::vector<uint32_t> code;   // per aux after the preimage: bit position, or 0x100 | mask index
    std::vector<uint32_t> masks;  // 8 words per mask
};
template <class P>
static BitDecompProgram make_bit_decomp_program() {
    BitDecompProgram bp;
    uint32_t pm1[8];
    for (int i = 0; i < 8; i++) pm1[i] = P::mod(i);
    pm1[0] -= 1;  // p is odd
    auto bit = [&](int i) { return (pm1[i >> 5] >> (i & 31)) & 1u; };
    uint32_t last[8] = {0}, cur[8];
    bool have_last = false, found = false;
    std::vector<int> run;
    for (int i = 255; i >= 0; i--) {
        const bool b = bit(i);
        found = found || b;
        if (!found) continue;
        if (b) {
            bp.code.push_back((uint32_t)i);
            run.push_back(i);
            continue;
        }
        if (!run.empty()) {
            for (int k = 0; k < 8; k++) cur[k] = 0;
            cur[run[0] >> 5] |= 1u << (run[0] & 31);
            auto emit_and = [&] {
                bp.code.push_back(0x100u | (uint32_t)(bp.masks.size() / 8));
                bp.masks.insert(bp.masks.end(), cur, cur + 8);
            };
            for (size_t m = 1; m < run.size(); m++) {
                cur[run[m] >> 5] |= 1u << (run[m] & 31);
                emit_and();
            }
            if (have_last) {
                for (int k = 0; k < 8; k++) cur[k] |= last[k];
                emit_and();
            }
            for (int k = 0; k < 8; k++) last[k] = cur[k];
            have_last = true;
            run.clear();
        }
        bp.code.push_back((uint32_t)i);
    }
    return bp;
}

template <class P>
__global__ __launch_bounds__(128) void bit_decomp_trace_kernel(const Fe<P>* __restrict__ vals, size_t n, const uint32_t* __restrict__ code, int ncode,
                                                                 const uint32_t* __restrict__ masks, int in_mont, TraceDst dst) {
    __shared__ uint32_t v[8];
    Fe<P>* w = (Fe<P>*)dst.w;
    const Fe<P> one = fe_one<P>(), zero = fe_zero<P>();
    for (size_t i = blockIdx.x; i < n; i += gridDim.x) {
        const size_t wb = dst.base(i);
        if (threadIdx.x == 0) {
            const Fe<P> x = vals[i];
            const Fe<P> c = in_mont ? fe_from_mont<P>(x) : x;
            for (int k = 0; k < 8; k++) v[k] = c.l[k];
            w[wb] = in_mont ? x : fe_to_mont<P>(x);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < ncode; j += 128) {
            const uint32_t cd = code[j];
            bool on;
            if (cd & 0x100u) {
                const uint32_t* m = masks + (size_t)(cd & 0xffu) * 8;
                on = true;
                for (int k = 0; k < 8; k++) on = on && ((v[k] & m[k]) == m[k]);
            } else {
                on = (v[cd >> 5] >> (cd & 31)) & 1u;
            }
            w[wb + 1 + j] = on ? one : zero;
        }
        __syncthreads();
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
    std::vector<uint32_t> post_words;  // neptune's post-S-box keys (slot-witness trace kernels), after the 2^522 record
    for (auto& x : neptune_post_keys<P>(pp))
        for (int i = 0; i < 8; i++) post_words.push_back(x.l[i]);
    pc->image = poseidon29_image<P>(poseidon_device_image<P>(pp), post_words);  // radix-2^29 form, 12 words per constant
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

// ---- slot witnesses: launch side -------------------------------------------------------------------------------------
template <class P, int T>
static void launch_trace(const void* d_pre, size_t n, PoseidonConsts& pc, int flags, const TraceDst& dst, hipStream_t s) {
    if (n == 0) return;
    const uint4* img = device_image(pc, s);
    const int img_vec4 = (int)(pc.image.size() / 4);
    const size_t lds_bytes = (size_t)img_vec4 * 16;
    ProfScope ps("poseidon_trace", s);
    if (n <= POSEIDON_WIDE_MAX) {
        auto wide = poseidon_trace_wide_kernel<P, T>;
        allow_dynamic_lds((const void*)wide, (int)lds_bytes);
        constexpr size_t per_block = (size_t)(POSEIDON_WIDE_BLOCK / 64) * (64 / T);
        hipLaunchKernelGGL(wide, dim3(div_up(n, per_block)), dim3(POSEIDON_WIDE_BLOCK), lds_bytes, s, (const uint4*)d_pre, n, img, img_vec4, pc.rf,
                           pc.rp, flags, dst);
    } else {
        auto kern = poseidon_trace_batch_kernel<P, T>;
        allow_dynamic_lds((const void*)kern, (int)lds_bytes);
        unsigned blocks = div_up(n, POSEIDON_BLOCK), cap = (unsigned)num_cus() * 2;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(POSEIDON_BLOCK), lds_bytes, s, (const uint4*)d_pre, n, img, img_vec4, pc.rf, pc.rp, flags, dst);
    }
    LURK_HIP_CHECK(hipGetLastError());
}
template <class P>
static void launch_trace_f(int arity, const void* d_pre, size_t n, PoseidonConsts& pc, int flags, const TraceDst& dst, hipStream_t s) {
    switch (arity) {
        case 3: launch_trace<P, 4>(d_pre, n, pc, flags, dst, s); break;
        case 4: launch_trace<P, 5>(d_pre, n, pc, flags, dst, s); break;
        case 6: launch_trace<P, 7>(d_pre, n, pc, flags, dst, s); break;
        case 8: launch_trace<P, 9>(d_pre, n, pc, flags, dst, s); break;
    }
}

struct BitDecompDev {
    BitDecompProgram host;
    std::map<int, std::pair<DevBuf, DevBuf>> dev;  // device id -> (code, masks)
};
static BitDecompDev& bit_decomp_program(int field_id) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<BitDecompDev>> progs;
    LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
    std::lock_guard<std::mutex> lk(mu);
    auto it = progs.find(field_id);
    if (it == progs.end()) {
        auto bd = std::make_unique<BitDecompDev>();
        bd->host = field_id == 0 ? make_bit_decomp_program<PallasFp>() : field_id == 1 ? make_bit_decomp_program<PallasFq>() : make_bit_decomp_program<Bn254Fr>();
        it = progs.emplace(field_id, std::move(bd)).first;
    }
    return *it->second;
}
template <class P>
static void launch_bit_decomp(BitDecompDev& bd, const void* d_vals, size_t n, int in_mont, const TraceDst& dst, hipStream_t s) {
    if (n == 0) return;
    const int dev = current_device();
    std::pair<DevBuf, DevBuf>* bufs;
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        auto it = bd.dev.find(dev);
        if (it == bd.dev.end()) {
            DevBuf code(bd.host.code.size() * 4), masks(bd.host.masks.size() * 4 + 32);
            LURK_HIP_CHECK(hipMemcpy(code.p, bd.host.code.data(), bd.host.code.size() * 4, hipMemcpyHostToDevice));
            if (!bd.host.masks.empty()) LURK_HIP_CHECK(hipMemcpy(masks.p, bd.host.masks.data(), bd.host.masks.size() * 4, hipMemcpyHostToDevice));
            it = bd.dev.emplace(dev, std::make_pair(std::move(code), std::move(masks))).first;
        }
        bufs = &it->second;
    }
    ProfScope ps("bit_decomp_trace", s);
    unsigned blocks = n < 65535 ? (unsigned)n : 65535u;
    hipLaunchKernelGGL((bit_decomp_trace_kernel<P>), dim3(blocks), dim3(128), 0, s, (const Fe<P>*)d_vals, n, bufs->first.as<uint32_t>(),
                       (int)bd.host.code.size(), bufs->second.as<uint32_t>(), in_mont, dst);
    LURK_HIP_CHECK(hipGetLastError());
}

static bool slot_is_hash(int slot_type) { return slot_type == 3 || slot_type == 4 || slot_type == 6 || slot_type == 8; }

size_t slot_witness_size(int field_id, int slot_type) {
    LURK_REQUIRE(slot_is_hash(slot_type) || slot_type == LURK_SLOT_BIT_DECOMP, "unknown slot type");
    if (slot_type == LURK_SLOT_BIT_DECOMP) return 1 + bit_decomp_program(field_id).host.code.size();
    PoseidonConsts& pc = get_consts(field_id, slot_type);
    return (size_t)slot_type + 3 * ((size_t)pc.t * pc.rf + pc.rp) + 1;
}

void slot_witness_device(int field_id, int slot_type, const void* d_pre, size_t n, int pre_mont, const TraceDst& dst, hipStream_t s) {
    LURK_REQUIRE(slot_is_hash(slot_type) || slot_type == LURK_SLOT_BIT_DECOMP, "unknown slot type");
    if (slot_type == LURK_SLOT_BIT_DECOMP) {
        BitDecompDev& bd = bit_decomp_program(field_id);
        if (field_id == 0) launch_bit_decomp<PallasFp>(bd, d_pre, n, pre_mont, dst, s);
        else if (field_id == 1) launch_bit_decomp<PallasFq>(bd, d_pre, n, pre_mont, dst, s);
        else launch_bit_decomp<Bn254Fr>(bd, d_pre, n, pre_mont, dst, s);
        return;
    }
    PoseidonConsts& pc = get_consts(field_id, slot_type);
    const int flags = pre_mont ? PF_IN_MONT : 0;
    if (field_id == 0) launch_trace_f<PallasFp>(slot_type, d_pre, n, pc, flags, dst, s);
    else if (field_id == 1) launch_trace_f<PallasFq>(slot_type, d_pre, n, pc, flags, dst, s);
    else launch_trace_f<Bn254Fr>(slot_type, d_pre, n, pc, flags, dst, s);
}

// All slot blocks of a MultiFrame in one call: the (up to five) launches are independent and, at a folding step's sizes,
// latency-bound (a few hundred waves each), so they run side by side on per-device helper streams forked from and joined
// back into the caller's stream.
struct TraceStreams {
    hipStream_t s[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};
static TraceStreams& trace_streams() {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<TraceStreams>> per_dev;
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_dev.find(dev);
    if (it == per_dev.end()) {
        auto ts = std::make_unique<TraceStreams>();
        LURK_HIP_CHECK(hipEventCreateWithFlags(&ts->fork, hipEventDisableTiming));
        for (int k = 0; k < 5; k++) {
            LURK_HIP_CHECK(hipStreamCreateWithFlags(&ts->s[k], hipStreamNonBlocking));
            LURK_HIP_CHECK(hipEventCreateWithFlags(&ts->join[k], hipEventDisableTiming));
        }
        it = per_dev.emplace(dev, std::move(ts)).first;
    }
    return *it->second;
}
static std::mutex g_frames_mu;  // the helper streams' events are shared: one frames call at a time per process

void frames_witness_device(int field_id, size_t num_frames, const size_t* counts, const void* const* d_pre, int pre_mont, void* d_w, size_t first,
                           size_t frame_len, hipStream_t s) {
    static const int types[5] = {LURK_SLOT_HASH4, LURK_SLOT_HASH6, LURK_SLOT_HASH8, LURK_SLOT_COMMITMENT, LURK_SLOT_BIT_DECOMP};
    size_t off = 0, sizes[5];
    for (int k = 0; k < 5; k++) {
        sizes[k] = counts[k] ? slot_witness_size(field_id, types[k]) : 0;
        LURK_REQUIRE(counts[k] == 0 || d_pre[k], "null preimage array for a slot type with a non-zero count");
        off += counts[k] * sizes[k];
    }
    LURK_REQUIRE(num_frames <= 1 || frame_len >= off, "frame_len shorter than the frame's slot blocks");
    if (num_frames == 0) return;
    std::lock_guard<std::mutex> lk(g_frames_mu);
    TraceStreams& ts = trace_streams();
    LURK_HIP_CHECK(hipEventRecord(ts.fork, s));
    off = 0;
    for (int k = 0; k < 5; k++) {
        if (!counts[k]) continue;
        LURK_HIP_CHECK(hipStreamWaitEvent(ts.s[k], ts.fork, 0));
        TraceDst dst{d_w, nullptr, first + off, sizes[k], counts[k], frame_len};
        slot_witness_device(field_id, types[k], d_pre[k], num_frames * counts[k], pre_mont, dst, ts.s[k]);
        LURK_HIP_CHECK(hipEventRecord(ts.join[k], ts.s[k]));
        LURK_HIP_CHECK(hipStreamWaitEvent(s, ts.join[k], 0));
        off += counts[k] * sizes[k];
    }
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

int lurk_hip_slot_witness_size(int field_id, int slot_type, size_t* size) {
    // pure host computation (the CPU tests pin it against the reference's constants)
    try {
        LURK_REQUIRE(size, "null output");
        *size = slot_witness_size(field_id, slot_type);
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}

int lurk_hip_slot_witness_dev(int field_id, int slot_type, const void* d_preimages, size_t n, int preimages_mont, void* d_w,
                              const uint64_t* d_offsets, size_t first, size_t stride, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(n == 0 || (d_preimages && d_w), "null buffer");
        LURK_REQUIRE(d_offsets || n <= 1 || stride >= slot_witness_size(field_id, slot_type), "stride shorter than a slot's block");
        TraceDst dst{d_w, d_offsets, first, stride};
        slot_witness_device(field_id, slot_type, d_preimages, n, preimages_mont, dst, (hipStream_t)stream);
    });
}

int lurk_hip_slot_witness(int field_id, int slot_type, const void* preimages, size_t n, int preimages_mont, void* w_out) {
    return guarded([&] {
        const size_t sz = slot_witness_size(field_id, slot_type);  // validates the arguments
        if (n == 0) return;
        LURK_REQUIRE(preimages && w_out, "null buffer");
        const size_t pre_elems = slot_type == LURK_SLOT_BIT_DECOMP ? 1 : (size_t)slot_type;
        DevBuf in(n * pre_elems * 32), out(n * sz * 32);
        LURK_HIP_CHECK(hipMemcpy(in.p, preimages, n * pre_elems * 32, hipMemcpyHostToDevice));
        TraceDst dst{out.p, nullptr, 0, sz};
        slot_witness_device(field_id, slot_type, in.p, n, preimages_mont, dst, nullptr);
        LURK_HIP_CHECK(hipStreamSynchronize(nullptr));
        LURK_HIP_CHECK(hipMemcpy(w_out, out.p, n * sz * 32, hipMemcpyDeviceToHost));
    });
}

int lurk_hip_frames_witness_dev(int field_id, size_t num_frames, const size_t* counts5, const void* const* d_preimages5, int preimages_mont,
                                void* d_w, size_t first, size_t frame_len, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(counts5 && d_preimages5 && (num_frames == 0 || d_w), "null argument");
        frames_witness_device(field_id, num_frames, counts5, d_preimages5, preimages_mont, d_w, first, frame_len, (hipStream_t)stream);
    });
}

int lurk_hip_witness_blocks_dev(void* d_w, size_t first, size_t stride, const void* src, int src_on_host, size_t n_blocks, size_t block_len,
                                void* stream) {
    return guarded([&] {
        if (n_blocks == 0 || block_len == 0) return;
        LURK_REQUIRE(d_w && src, "null buffer");
        LURK_REQUIRE(n_blocks == 1 || stride >= block_len, "blocks overlap");
        LURK_HIP_CHECK(hipMemcpy2DAsync((char*)d_w + first * 32, stride * 32, src, block_len * 32, block_len * 32, n_blocks,
                                        src_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, (hipStream_t)stream));
    });
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

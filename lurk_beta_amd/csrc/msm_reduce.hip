// msm_reduce.hip - the bucket reduction of the Pippenger pipeline (msm.hip, stage 6).
//
// sum_b b * B_b with bucket b at index b - 1:  sum (idx + 1) X_idx = S + sum_k 2^k P_k,  S = sum X,  P_k = sum of the X whose idx has
// bit k set.  The (S, P_0 .. P_{k-1}) vectors of two adjacent segments of 2^k buckets merge with k + 1 independent additions (the new
// plane k is the upper half's S), so the reduction is c - 1 levels of depth ONE addition: a running-sum formulation needs > 100
// dependent additions, and on this chip a dependent kernel boundary costs ~1.5 us while a dependent addition costs 5-7 us.
//
// One launch per level (c - 1 = 19 at 20-bit windows).  What round 3 changed:
//   * points travel between levels on the radix-2^29 layer (curve29.cuh: xyzz29_add, the inlined 135-mad multiplier - no call per
//     product, no conversion between levels): 160-byte records with the identity as a flag; level 0 converts the buckets on the fly,
//     the last level stores ordinary XYZZ points straight into the slot's pinned host buffer (no copy node behind it);
//   * measured and dropped: ALL levels in one persistent launch behind grid barriers (512 workgroups, one monotonic counter, agent-
//     scope release / relaxed poll / acquire): 1.07 ms per reduction against 0.45 for the launches - a counter barrier costs 13 us
//     at two workgroups per CU (MI355X_MICROARCH.md, price list: barrier-counter) and its release fence writes the L2 back, a kernel
//     boundary 1.5 us.  The guide's verdict for all-to-all seams - cut the launch there - holds for this one too;
//   * measured and dropped in round 4: the last SIX levels (64 segments per key space from level c - 7 on) in one launch, one workgroup
//     per key space merging through LDS with a barrier per level: 0.339 / 0.390 ms per reduction at 2^20 / 2^22 against 0.343 / 0.384
//     with one launch per level (profiles/r04_run3_sweep.jsonl) - a level costs its ONE dependent XYZZ addition (~3 000 instructions of
//     4+ cycles on a single wave: 6 us) whether a kernel boundary follows it or a barrier;
//   * measured and dropped in round 4: a component-major thread order (the copied component of every pair in waves of its own, so that
//     level 0 runs half the adding waves, level 1 two thirds): 0.341 / 0.373 ms per reduction at 2^20 / 2^22 against 0.343 / 0.379 -
//     the early levels are not bound by the additions of their idle lanes either;
//   * round 4, adopted: the kernel is held to 128 VGPRs (__launch_bounds__(256, 4): 11 of its 183 registers spill).  With 183 a level's
//     workgroups could not be placed on a SIMD beside two resident one-wave accumulations (2 x 176 + 184 > 512): with commitments in
//     flight every one of the 19 dependent levels waited for an accumulation to END - reductions of 4-6 ms in the device-side timeline,
//     0.7-1.4 ms now - and the slot came back that much later.  2^22, three in flight: 953-961 -> 969-972 Mscalar-mul/s on one box
//     (alternating runs), 2^20, two in flight: 790 -> 812; alone the reduction got 4 % faster too (four waves per SIMD in the early
//     levels): profiles/r04_msm_pipeline_coresidency.txt.
//   * round 6, adopted: the LATE levels as wave butterflies (msm_planes29_wave_kernel).  From the level where a whole level is fewer
//     waves than the chip has SIMDs, a level's launch costs ~10 us for its one 4.5 us addition (launch, two dependent 160-byte
//     loads, the store: rocprofv3 trace of a lone 2^20 commitment, profiles/r06_reduce_levels.txt: 13 launches of 7-14 us).  A
//     component of the merged vector is a plain SUM over the merged segments (the new planes: a sum of the S values of the segments
//     whose index has that bit set), so one wave takes 64 adjacent segments of ONE component and sums them with six
//     exchange-and-add steps across its lanes: six levels per launch, nothing between them but 37 ds_bpermute.  It does ~3x the
//     additions of the tree (lanes idle from the second step on) - which is why the first levels stay a tree.
//   * round 6, adopted: the EARLY levels in pairs (msm_planes29_quad_kernel).  With their additions replaced by an xor the seven early
//     launches take 191 us against 203: they are bound by their memory side (160-byte records, one per lane), not by the VALU - which is
//     why the component-major order of round 4 gained nothing (re-measured: - 7..14 us, kept as the adding-threads-first order).  A pair
//     of levels that never writes the level between them moves half the bytes: levels 0-6 of a 2^20 commitment 203 -> 155 us
//     (63 + 49 + 28 for the pairs, 16 for level 6), the reduction 0.286 -> 0.245 ms, a 2^20 commitment 1.41 -> 1.37 ms synchronous,
//     1.18 -> 1.15 with two in flight, the folding step 3.01 -> 2.93 ms (one box, alternating: profiles/r06_reduce_levels.txt).
#include "common.hpp"
#include "msm_core.cuh"
#include "curve29.cuh"

// register budget of the reduction kernels: 128 VGPRs (four waves per SIMD; see "round 4, adopted" above) unless a build overrides it
#ifdef LURK_REDUCE_VGPRS  // here: waves per SIMD the compiler is to leave room for (4 -> 128 VGPRs, 3 -> 168, 2 -> 256)
#define LURK_REDUCE_BOUNDS __launch_bounds__(REDUCE_BLOCK, LURK_REDUCE_VGPRS)  /* (the second argument: workgroups per CU = waves per SIMD at 256 threads) */
#else
#define LURK_REDUCE_BOUNDS __launch_bounds__(REDUCE_BLOCK, 4)
#endif
#ifndef LURK_REDUCE_COMPACT
#define LURK_REDUCE_COMPACT 1
#endif

namespace lurk {

#if defined(LURK_REDUCE_NOADD)  // experiment: the levels' memory side alone (results are wrong)
template <class P>
__device__ __forceinline__ void level_add(Xyzz29<P>& r, bool& r_id, const Xyzz29<P>& h, bool h_id) {
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.l[i] ^= h.x.l[i]; r.y.l[i] ^= h.y.l[i]; r.zz.l[i] ^= h.zz.l[i]; r.zzz.l[i] ^= h.zzz.l[i]; }
    r_id = r_id && h_id;
}
#else
template <class P>
__device__ __forceinline__ void level_add(Xyzz29<P>& r, bool& r_id, const Xyzz29<P>& h, bool h_id) { xyzz29_add<P>(r, r_id, h, h_id); }
#endif

constexpr int REDUCE_BLOCK = 256;

template <class P>
struct Plane29 {
    Xyzz29<P> p;
    uint32_t id, pad[3];
};

template <class P>
__device__ __forceinline__ void plane_load(const Plane29<P>* __restrict__ src, Xyzz29<P>& v, bool& id) {
    const Plane29<P> r = *src;
    v = r.p;
    id = r.id != 0;
}

// level k: in[g][seg][0..k] (segments of 2^k buckets; level 0 reads the buckets themselves) -> out[g][seg / 2][0..k+1]
// LAST: the G x c plane sums leave as ordinary XYZZ points (out_host, pinned host memory) instead of plane records
template <class P, bool FIRST, bool LAST>
__global__ LURK_REDUCE_BOUNDS void msm_planes29_kernel(const Xyzz<P>* __restrict__ buckets, const Plane29<P>* __restrict__ in,
                                                                      Plane29<P>* __restrict__ out, Xyzz<P>* __restrict__ out_host, int k, int G, uint32_t B) {
    __builtin_amdgcn_s_setprio(3);  // a latency-bound tail kernel: issue ahead of an accumulation sharing the SIMD
    const size_t nseg_out = (size_t)B >> (k + 1);
    const size_t comps_out = (size_t)k + 2, comps_in = (size_t)k + 1;
    const size_t tid = (size_t)blockIdx.x * REDUCE_BLOCK + threadIdx.x;
    if (tid >= (size_t)G * nseg_out * comps_out) return;
#if LURK_REDUCE_COMPACT
    // the adding threads first (k + 1 components per output segment), the copying ones (the new plane) in waves of their own behind them
    size_t comp, seg, g;
    const size_t adders = (size_t)G * nseg_out * comps_in;
    if (tid < adders) {
        comp = tid % comps_in;
        seg = (tid / comps_in) % nseg_out;
        g = tid / (comps_in * nseg_out);
    } else {
        comp = comps_in;
        seg = (tid - adders) % nseg_out;
        g = (tid - adders) / nseg_out;
    }
    const size_t id = (g * nseg_out + seg) * comps_out + comp;
#else
    const size_t id = tid;
    const size_t comp = id % comps_out, seg = (id / comps_out) % nseg_out, g = id / (comps_out * nseg_out);
#endif
    Xyzz29<P> r, h;
    bool r_id, h_id;
    if (FIRST) {  // segments of one bucket, one component (S)
        const Xyzz<P> hi = buckets[g * B + 2 * seg + 1];
        xyzz29_from_xyzz<P>(hi, h, h_id);
        if (comp == 0) {
            const Xyzz<P> lo = buckets[g * B + 2 * seg];
            xyzz29_from_xyzz<P>(lo, r, r_id);
            level_add<P>(r, r_id, h, h_id);
        } else {
            r = h;
            r_id = h_id;
        }
    } else {
        const Plane29<P>* lo = in + ((g * (nseg_out * 2) + 2 * seg) * comps_in);
        const Plane29<P>* hi = lo + comps_in;
        if (comp <= (size_t)k) {
            plane_load<P>(lo + comp, r, r_id);
            plane_load<P>(hi + comp, h, h_id);
            level_add<P>(r, r_id, h, h_id);
        } else {
            plane_load<P>(hi, r, r_id);  // the new plane k: the upper half's S
        }
    }
    if (LAST) {
        out_host[id] = xyzz29_to_xyzz<P>(r, r_id);
    } else {
        Plane29<P> o;
        o.p = r;
        o.id = r_id;
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        out[id] = o;
    }
}

// Levels k and k + 1 in one launch (round 6): the early levels are bound by their memory side, not by their additions (with the additions
// replaced by an xor the seven level launches of a 2^20 commitment take 191 us against 203: profiles/r06_reduce_levels.txt), so a pair of
// levels that never writes the level between them halves the traffic of the pair.  Four adjacent segments of 2^k buckets (k + 1
// components each) merge into one of 2^(k+2) (k + 3 components) with exactly the tree's additions: a component below k + 1 is the sum of
// the four (three additions, one thread: in2 + in3, + in0, + in1), the new plane k is S1 + S3 (one addition, a thread of its own), the
// new plane k + 1 is S2 + S3 - what the S thread holds after its first addition, stored from there.  The three-addition threads come
// first, the one-addition threads in waves of their own behind them.  FIRST: the inputs are the buckets themselves (k = 0).
template <class P, bool FIRST>
__global__ LURK_REDUCE_BOUNDS void msm_planes29_quad_kernel(const Xyzz<P>* __restrict__ buckets, const Plane29<P>* __restrict__ in,
                                                                           Plane29<P>* __restrict__ out, int k, int G, uint32_t B) {
    __builtin_amdgcn_s_setprio(3);
    const size_t nseg_out = (size_t)B >> (k + 2);
    const size_t ci = (size_t)k + 1, co = ci + 2;
    const size_t tid = (size_t)blockIdx.x * REDUCE_BLOCK + threadIdx.x;
    const size_t heavy = (size_t)G * nseg_out * ci;
    if (tid >= heavy + (size_t)G * nseg_out) return;
    size_t comp, seg, g;
    const bool light = tid >= heavy;
    if (!light) {
        comp = tid % ci;
        seg = (tid / ci) % nseg_out;
        g = tid / (ci * nseg_out);
    } else {
        comp = 0;  // (reads S)
        seg = (tid - heavy) % nseg_out;
        g = (tid - heavy) / nseg_out;
    }
    auto load = [&](int q, Xyzz29<P>& v, bool& id) {
        if (FIRST) {
            const Xyzz<P> b = buckets[g * B + 4 * seg + q];
            xyzz29_from_xyzz<P>(b, v, id);
        } else {
            plane_load<P>(in + ((g * (nseg_out * 4) + 4 * seg + q) * ci + comp), v, id);
        }
    };
    auto store = [&](size_t c_out, const Xyzz29<P>& v, bool id) {
        Plane29<P> o;
        o.p = v;
        o.id = id;
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        out[(g * nseg_out + seg) * co + c_out] = o;
    };
    Xyzz29<P> r, h;
    bool r_id, h_id;
    if (light) {
        load(1, r, r_id);
        load(3, h, h_id);
        level_add<P>(r, r_id, h, h_id);
        store(ci, r, r_id);
        return;
    }
    load(2, r, r_id);
    load(3, h, h_id);
    level_add<P>(r, r_id, h, h_id);
    if (comp == 0) store(ci + 1, r, r_id);
    load(0, h, h_id);
    level_add<P>(r, r_id, h, h_id);
    load(1, h, h_id);
    level_add<P>(r, r_id, h, h_id);
    store(comp, r, r_id);
}

// LOG consecutive levels in one launch, from the input of level K (segments of 2^K buckets, K + 1 components each, nseg_in = B >> K of
// them per key space): one wave per (key space, output component, 64 adjacent input segments); lane l holds segment l's record of the
// component - for a new plane K + t, segment l's S if bit t of l is set, the identity if not - and LOG exchange-and-add steps leave the
// sum of every 2^LOG adjacent segments in the group's first lane.  Output layout = the level kernel's (out[g][seg >> LOG][0 .. K + LOG]).
template <class P, bool LAST>
__global__ LURK_REDUCE_BOUNDS void msm_planes29_wave_kernel(const Plane29<P>* __restrict__ in, Plane29<P>* __restrict__ out,
                                                                           Xyzz<P>* __restrict__ out_host, int K, int LOG, int G, uint32_t nseg_in) {
    __builtin_amdgcn_s_setprio(3);
    const size_t comps_in = (size_t)K + 1, comps_out = comps_in + LOG;
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * REDUCE_BLOCK + threadIdx.x) >> 6;
    const size_t chunks = ((size_t)nseg_in + 63) / 64;
    if (wave >= (size_t)G * comps_out * chunks) return;
    const size_t chunk = wave % chunks, comp = (wave / chunks) % comps_out, g = wave / (chunks * comps_out);
    const size_t seg = chunk * 64 + lane;
    Xyzz29<P> r;
    bool r_id = true;
#pragma unroll
    for (int i = 0; i < 9; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
    if (seg < nseg_in) {
        const Plane29<P>* src = in + (g * nseg_in + seg) * comps_in;
        if (comp < comps_in) plane_load<P>(src + comp, r, r_id);
        else if ((seg >> (comp - comps_in)) & 1) plane_load<P>(src, r, r_id);
    }
    for (int t = 0; t < LOG; t++) {
        const int m = 1 << t;
        Xyzz29<P> h;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            h.x.l[i] = (uint32_t)__shfl_xor((int)r.x.l[i], m);
            h.y.l[i] = (uint32_t)__shfl_xor((int)r.y.l[i], m);
            h.zz.l[i] = (uint32_t)__shfl_xor((int)r.zz.l[i], m);
            h.zzz.l[i] = (uint32_t)__shfl_xor((int)r.zzz.l[i], m);
        }
        const bool h_id = __shfl_xor((int)r_id, m) != 0;
        xyzz29_add<P>(r, r_id, h, h_id);  // (both lanes of a pair form the same sum; only the lower one's is used from here on)
    }
    if ((lane & ((1u << LOG) - 1u)) != 0 || seg >= nseg_in) return;
    const size_t nseg_out = (size_t)nseg_in >> LOG;
    const size_t id = (g * nseg_out + (seg >> LOG)) * comps_out + comp;
    if (LAST) {
        out_host[id] = xyzz29_to_xyzz<P>(r, r_id);
    } else {
        Plane29<P> o;
        o.p = r;
        o.id = r_id;
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        out[id] = o;
    }
}

size_t msm_reduce_plane_bytes(size_t nb) { return nb * 160; }

// LURK_MSM_REDUCE_WAVE=0: every level as its own launch (the round-3 form)
static bool reduce_wave_levels() {
    static const bool on = [] { const char* v = getenv("LURK_MSM_REDUCE_WAVE"); return !v || atoi(v) != 0; }();
    return on;
}

// LURK_MSM_REDUCE_QUAD=0: no level pairs
static bool reduce_quad_levels() {
    static const bool on = [] { const char* v = getenv("LURK_MSM_REDUCE_QUAD"); return !v || atoi(v) != 0; }();
    return on;
}

// planes_a / planes_b: msm_reduce_plane_bytes(G * B) each; out_host: G * c XYZZ points of pinned host memory
template <class P>
void msm_launch_reduce(const Xyzz<P>* buckets, void* planes_a, void* planes_b, int c, int G, uint32_t B, Xyzz<P>* out_host, hipStream_t s) {
    static_assert(sizeof(Plane29<P>) == 160, "plane record");
    Plane29<P>* bufs[2] = {(Plane29<P>*)planes_a, (Plane29<P>*)planes_b};
    // the last <= 12 levels as two wave butterflies, starting where the first of them is at most two waves per SIMD
    int K = c - 1;  // levels [0, K) one launch each
    if (reduce_wave_levels() && c >= 3 && B == (1u << (c - 1))) {
        K = c - 1 > 12 ? c - 1 - 12 : 0;
        if (K < 1) K = 1;  // (level 0 converts the buckets)
        const size_t cap = (size_t)num_cus() * 4 * 2;
        auto waves = [&](int k) {
            const int rem = c - 1 - k, log1 = rem > 6 ? rem - rem / 2 : rem;
            return (size_t)G * ((((size_t)B >> k) + 63) / 64) * (size_t)(k + 1 + log1);
        };
        while (K < c - 1 && waves(K) > cap) K++;
    }
    // levels [0, K): in pairs that skip the level between them (LURK_MSM_REDUCE_QUAD=0: one launch each), an odd one out as it was
    const Plane29<P>* src = bufs[1];
    Plane29<P>* dst = bufs[0];
    auto swap = [&] {
        Plane29<P>* t = dst;
        dst = const_cast<Plane29<P>*>(src);
        src = t;
    };
    int k = 0;
    while (k < K) {
        if (reduce_quad_levels() && k + 2 <= K && k + 2 <= c - 2) {
            const size_t threads = (size_t)G * ((size_t)B >> (k + 2)) * (k + 2);
            const dim3 grid(div_up(threads, REDUCE_BLOCK)), block(REDUCE_BLOCK);
            if (k == 0) hipLaunchKernelGGL((msm_planes29_quad_kernel<P, true>), grid, block, 0, s, buckets, src, dst, k, G, B);
            else hipLaunchKernelGGL((msm_planes29_quad_kernel<P, false>), grid, block, 0, s, buckets, src, dst, k, G, B);
            k += 2;
        } else {
            const size_t threads = (size_t)G * ((size_t)B >> (k + 1)) * (k + 2);
            const dim3 grid(div_up(threads, REDUCE_BLOCK)), block(REDUCE_BLOCK);
            const bool last = k == c - 2;
            if (k == 0 && last) hipLaunchKernelGGL((msm_planes29_kernel<P, true, true>), grid, block, 0, s, buckets, src, dst, out_host, k, G, B);
            else if (k == 0) hipLaunchKernelGGL((msm_planes29_kernel<P, true, false>), grid, block, 0, s, buckets, src, dst, out_host, k, G, B);
            else if (last) hipLaunchKernelGGL((msm_planes29_kernel<P, false, true>), grid, block, 0, s, buckets, src, dst, out_host, k, G, B);
            else hipLaunchKernelGGL((msm_planes29_kernel<P, false, false>), grid, block, 0, s, buckets, src, dst, out_host, k, G, B);
            k += 1;
        }
        swap();  // what was written is the next launch's input
    }
    while (k < c - 1) {
        const int rem = c - 1 - k, log = rem > 6 ? rem - rem / 2 : rem;
        const uint32_t nseg_in = B >> k;
        const size_t nwaves = (size_t)G * (((size_t)nseg_in + 63) / 64) * (size_t)(k + 1 + log);
        const dim3 grid(div_up(nwaves * 64, REDUCE_BLOCK)), block(REDUCE_BLOCK);
        if (k + log == c - 1) hipLaunchKernelGGL((msm_planes29_wave_kernel<P, true>), grid, block, 0, s, src, dst, out_host, k, log, G, nseg_in);
        else hipLaunchKernelGGL((msm_planes29_wave_kernel<P, false>), grid, block, 0, s, src, dst, out_host, k, log, G, nseg_in);
        swap();
        k += log;
    }
    LURK_HIP_CHECK(hipGetLastError());
}
template void msm_launch_reduce<PallasFp>(const Xyzz<PallasFp>*, void*, void*, int, int, uint32_t, Xyzz<PallasFp>*, hipStream_t);
template void msm_launch_reduce<PallasFq>(const Xyzz<PallasFq>*, void*, void*, int, int, uint32_t, Xyzz<PallasFq>*, hipStream_t);

}  // namespace lurk

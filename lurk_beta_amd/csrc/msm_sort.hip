// msm_sort.hip - the two-pass partitioned counting sort of a commitment's signed window digits (stage 2 of msm_core.cuh's
// pipeline; the caller is arecibo's CommitmentEngine::commit through msm.hip: /root/reference/src/proof/nova.rs:287-293).
// msm_sort.hpp describes the passes and what round 5 changed.
#include "msm_sort.hpp"

#include "common.hpp"
#include "msm_core.cuh"

namespace lurk {

MsmShape msm_make_shape(int c, bool precomputed, size_t npoints, size_t n, int sel) {
    MsmShape sh;
    sh.c = c;
    sh.W = msm_num_windows(c);
    sh.sel = precomputed ? sel : -1;
    sh.low_prio = 0;
    sh.G = precomputed ? (sh.sel >= 0 ? 2 : 1) : sh.W;
    sh.B = 1u << (c - 1);
    sh.NB = (uint32_t)sh.G * sh.B;
    // partitions: as few as let an average partition (W n / P entries) fit the LDS stage of pass 2 with 15 % to
    // spare - 2048 up to n = 2^22, 4096 for the rc = 900 step circuit (n ~ 10^7), 8192 beyond - and enough of them
    // that the low key bits of pass 2 fit one byte (a pair's two key spaces of 2^19: 4096)
    sh.P = MSM_P_MIN;
    while (sh.P < MSM_P_MAX && (uint32_t)sh.P < sh.NB) {
        int lb = 0;
        while ((sh.NB >> lb) > (uint32_t)sh.P) lb++;
        if (lb <= MSM_LB_MAX && (double)sh.W * (double)n / sh.P <= 0.85 * (double)msm_part2_cap(lb)) break;
        sh.P *= 2;
    }
    sh.LB = 0;
    while ((sh.NB >> sh.LB) > (uint32_t)sh.P) sh.LB++;
    LURK_REQUIRE(sh.LB <= MSM_LB_MAX, "sort shape: more than 8 low key bits per partition");
    sh.tile = MSM_SORT_BLOCK;
    while (sh.tile > 64 && msm_scatter1_lds(sh.P, sh.W, sh.tile) > MSM_LDS_BYTES) sh.tile /= 2;
    // Task length.  A task is a chain of dependent mixed additions (4.5 us each on a lane): 64-entry tasks suit commitments whose
    // W n / 64 tasks fill the chip's ~131 072 lanes; a 65 536-point commitment under a 16-bit window table has 2^20 entries - 16 384
    // tasks of 64 would leave seven lanes in eight idle behind chains of 32-64 additions.  Halve S until the tasks fill the lanes: the
    // per-bucket partials (<= 16 by one lane, more by a workgroup tree) absorb the rest.
    sh.S = MSM_S;
    static const size_t task_target = [] { const char* e = getenv("LURK_MSM_TASK_TARGET"); long v = e ? atol(e) : 0; return v > 0 ? (size_t)v : MSM_TASK_TARGET; }();
    while (sh.S > MSM_S_MIN && (size_t)sh.W * n / sh.S < task_target) sh.S /= 2;
    sh.NG = (int)(sh.NB / MSM_GRP);
    sh.n = n;
    sh.stride = precomputed ? npoints : 0;
    return sh;
}

// Every kernel of a commitment except the bucket accumulation is short and bound by latency, LDS atomics or HBM; with
// commitments in flight they share the SIMDs with the (older, VALU-saturating) accumulate waves of the previous
// commitment, and the instruction arbiter serves the oldest wave first: measured 15-18x slowdowns of these kernels.
// Raising their wave priority lets them issue when they are ready; they need a few percent of the VALU.
__device__ __forceinline__ void msm_sort_wave_prio(int low = 0) {
    if (low) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(3);
}

// entry (w, i) -> key = space * B + |d| - 1 (space = w in plain mode, 0 with the table)
__device__ __forceinline__ uint32_t msm_key(const MsmShape& sh, uint32_t w, uint32_t mag, size_t i) {
    const uint32_t space = sh.sel >= 0 ? (uint32_t)((i >> sh.sel) & 1u) : (sh.G == 1 ? 0u : w);
    return space * sh.B + mag - 1u;
}

constexpr int msm_ct_windows(int C) { return C ? (256 + C - 1) / C : 1; }

// ---- pass 1a: canonical scalars + coarse histogram ------------------------------------------------
// block blk owns scalars [blk*chunk, (blk+1)*chunk).  MONT: the scalars arrive in Montgomery form; the canonical form is stored
// for pass 1b (one read of the caller's vector per commitment).  C: window bits at compile time, 0 = sh.c (the register walk).
template <class SF, int C, bool MONT>
__global__ __launch_bounds__(MSM_SORT_BLOCK) void msm_hist1_kernel(const uint4* __restrict__ scalars, uint4* __restrict__ canon,
                                                                     uint32_t* __restrict__ block_hist, MsmShape sh, size_t chunk) {
    msm_sort_wave_prio(sh.low_prio);
    extern __shared__ uint32_t h[];  // [P]
    for (int p = threadIdx.x; p < sh.P; p += MSM_SORT_BLOCK) h[p] = 0;
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < sh.n ? lo + chunk : sh.n;
    // the next scalar is in flight while the current one is recoded (16 waves per CU do not hide the load otherwise)
    size_t i = lo + threadIdx.x;
    uint4 nlo = make_uint4(0, 0, 0, 0), nhi = nlo;
    if (i < hi) { nlo = scalars[2 * i]; nhi = scalars[2 * i + 1]; }
    for (; i < hi; i += MSM_SORT_BLOCK) {
        Fe<SF> s;
        s.l[0] = nlo.x; s.l[1] = nlo.y; s.l[2] = nlo.z; s.l[3] = nlo.w;
        s.l[4] = nhi.x; s.l[5] = nhi.y; s.l[6] = nhi.z; s.l[7] = nhi.w;
        if (i + MSM_SORT_BLOCK < hi) { nlo = scalars[2 * (i + MSM_SORT_BLOCK)]; nhi = scalars[2 * (i + MSM_SORT_BLOCK) + 1]; }
        if constexpr (MONT) {
            s = fe_from_mont<SF>(s);
            canon[2 * i] = make_uint4(s.l[0], s.l[1], s.l[2], s.l[3]);
            canon[2 * i + 1] = make_uint4(s.l[4], s.l[5], s.l[6], s.l[7]);
        }
        if constexpr (C != 0) {
            uint32_t d[msm_ct_windows(C)];
            msm_digits_ct<C>(s.l, d);
#pragma unroll
            for (int w = 0; w < msm_ct_windows(C); w++) {
                const uint32_t mag = d[w] & ~MSM_SIGN;
                if (mag) atomicAdd(&h[msm_key(sh, (uint32_t)w, mag, i) >> sh.LB], 1u);
            }
        } else {
            uint32_t carry = 0;
            uint32_t r[8];
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = s.l[k];
            for (int w = 0; w < sh.W; w++) {
                uint32_t mag = msm_digit_next(r, sh.c, carry) & ~MSM_SIGN;
                if (mag) atomicAdd(&h[msm_key(sh, (uint32_t)w, mag, i) >> sh.LB], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < sh.P; p += MSM_SORT_BLOCK) block_hist[(size_t)blockIdx.x * sh.P + p] = h[p];
}

// block p: exclusive scan of partition p's counts over the MSM_NB1 pass-1 blocks
__global__ __launch_bounds__(MSM_NB1) void msm_scan1_kernel(uint32_t* __restrict__ block_hist, uint32_t* __restrict__ part_cnt, int P) {
    msm_sort_wave_prio();
    __shared__ uint32_t sh[MSM_NB1];
    const int p = blockIdx.x, t = threadIdx.x;
    uint32_t v = block_hist[(size_t)t * P + p];
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < MSM_NB1; off <<= 1) {
        uint32_t a = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += a;
        __syncthreads();
    }
    block_hist[(size_t)t * P + p] = sh[t] - v;
    if (t == MSM_NB1 - 1) part_cnt[p] = sh[t];
}

// Exclusive scan over the 1024 threads of a sort block (wave scans + one scan of the 16 wave totals).
// scr: >= 17 words of LDS; returns the exclusive prefix of v, *total = sum over the block.
__device__ __forceinline__ uint32_t msm_block_scan(uint32_t v, uint32_t* scr, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) scr[wave] = inc;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t w = threadIdx.x < MSM_SORT_BLOCK / 64 ? scr[threadIdx.x] : 0, winc = w;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            uint32_t o = __shfl_up(winc, off);
            if (lane >= off) winc += o;
        }
        if (threadIdx.x < MSM_SORT_BLOCK / 64) scr[threadIdx.x] = winc - w;
        if (threadIdx.x == MSM_SORT_BLOCK / 64 - 1) scr[16] = winc;
    }
    __syncthreads();
    uint32_t ex = scr[wave] + inc - v;
    *total = scr[16];
    __syncthreads();  // scr may be reused at once
    return ex;
}

// single block: part_start[0..P] = exclusive scan of part_cnt
// (it also zeroes the small counters the later stages start from - the hot-bucket count, the task-length histogram, the persistent
// kernel's cursor block: three fill launches of ~6 us each in a chain of dependent launches otherwise)
__global__ __launch_bounds__(MSM_SORT_BLOCK) void msm_part_start_kernel(const uint32_t* __restrict__ part_cnt, uint32_t* __restrict__ part_start,
                                                                          int P, uint32_t* __restrict__ z0, int n0, uint32_t* __restrict__ z1, int n1,
                                                                          uint32_t* __restrict__ z2, int n2) {
    msm_sort_wave_prio();
    for (int i = threadIdx.x; i < n0; i += MSM_SORT_BLOCK) z0[i] = 0;
    for (int i = threadIdx.x; i < n1; i += MSM_SORT_BLOCK) z1[i] = 0;
    for (int i = threadIdx.x; i < n2; i += MSM_SORT_BLOCK) z2[i] = 0;
    __shared__ uint32_t scr[32];
    const int PER = P / MSM_SORT_BLOCK;  // 2, 4 or 8 counters per thread
    const int t = threadIdx.x;
    uint32_t c[MSM_P_PER_MAX], sum = 0;
#pragma unroll
    for (int j = 0; j < MSM_P_PER_MAX; j++) { c[j] = j < PER ? part_cnt[t * PER + j] : 0u; sum += c[j]; }
    uint32_t total;
    uint32_t run = msm_block_scan(sum, scr, &total);
#pragma unroll
    for (int j = 0; j < MSM_P_PER_MAX; j++)
        if (j < PER) { part_start[t * PER + j] = run; run += c[j]; }
    if (t == 0) part_start[P] = total;
}

// ---- pass 1b: scatter into the coarse partitions ---------------------------------------------------
// A tile of sh.tile (1024) canonical scalars yields <= W*tile entries; they are first grouped by partition in LDS (a block-local
// counting sort) and then copied out slot by slot, so that neighbouring lanes write neighbouring addresses: the entries of one
// partition leave as one run instead of as isolated stores (which cost a 32-byte sector each: rocprof showed 3.9x write
// amplification for the direct scatter).  Out: inter_e[slot] = table index | sign, inter_k[slot] = the key's low LB bits.
// C != 0: the tile's counting and placement sweeps share ONE recoding held in W registers.
#ifndef LURK_SORT_MIN_BLOCKS
#define LURK_SORT_MIN_BLOCKS 1  // (HIP: minimum WAVES per SIMD) 8: the compiler holds the kernel to 64 registers
#endif
template <int C>
__global__ __launch_bounds__(MSM_SORT_BLOCK, LURK_SORT_MIN_BLOCKS) void msm_scatter1_kernel(const uint4* __restrict__ scalars, const uint32_t* __restrict__ block_off,
                                                                        const uint32_t* __restrict__ part_start, uint32_t* __restrict__ inter_e,
                                                                        uint8_t* __restrict__ inter_k, MsmShape sh, size_t chunk) {
    msm_sort_wave_prio(sh.low_prio);
    extern __shared__ uint32_t lds[];
    const int P = sh.P, PER = P / MSM_SORT_BLOCK;
    uint32_t* goff = lds;          // [P] where this block's next entry of partition p goes
    uint32_t* cnt = goff + P;      // [P] entries of the current tile, then the placement cursor
    uint32_t* start = cnt + P;     // [P] exclusive scan of cnt
    uint32_t* scr = start + P;     // [32]
    uint2* stage = reinterpret_cast<uint2*>(scr + 32);  // [W * tile] (key, entry)
    const int t = threadIdx.x;
    const uint32_t low_mask = (1u << sh.LB) - 1u;
    for (int p = t; p < P; p += MSM_SORT_BLOCK) {
        goff[p] = part_start[p] + block_off[(size_t)blockIdx.x * P + p];
        cnt[p] = 0;
    }
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < sh.n ? lo + chunk : sh.n;
    for (size_t base = lo; base < hi; base += sh.tile) {
        const size_t i = base + t;
        const bool live = t < sh.tile && i < hi;
        uint32_t sl[8];
        uint32_t d[msm_ct_windows(C)];
        if (live) {
            // (keeping the NEXT tile's scalar in flight across the tile costs 8 registers per lane: 84 instead of 65, one allocation
            // granule too many for four sort waves to sit on a SIMD beside a resident accumulate wave (4 x 88 + 176 > 512) - the sort
            // of every commitment in flight then waited for an accumulation to END: 988 -> 839 Mscalar-mul/s at 2^22.  See the
            // register-fit test, tests/test_cabi_exports.py)
            const uint4 a = scalars[2 * i], b = scalars[2 * i + 1];
            sl[0] = a.x; sl[1] = a.y; sl[2] = a.z; sl[3] = a.w;
            sl[4] = b.x; sl[5] = b.y; sl[6] = b.z; sl[7] = b.w;
            if constexpr (C != 0) {
                msm_digits_ct<C>(sl, d);
#pragma unroll
                for (int w = 0; w < msm_ct_windows(C); w++) {
                    const uint32_t mag = d[w] & ~MSM_SIGN;
                    if (mag) atomicAdd(&cnt[msm_key(sh, (uint32_t)w, mag, i) >> sh.LB], 1u);
                }
            } else {
                uint32_t carry = 0;
                uint32_t r[8];
#pragma unroll
                for (int k = 0; k < 8; k++) r[k] = sl[k];
                for (int w = 0; w < sh.W; w++) {
                    uint32_t mag = msm_digit_next(r, sh.c, carry) & ~MSM_SIGN;
                    if (mag) atomicAdd(&cnt[msm_key(sh, (uint32_t)w, mag, i) >> sh.LB], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t total;
        {
            uint32_t c[MSM_P_PER_MAX], sum = 0;
#pragma unroll
            for (int j = 0; j < MSM_P_PER_MAX; j++) { c[j] = j < PER ? cnt[t * PER + j] : 0u; sum += c[j]; }
            uint32_t run = msm_block_scan(sum, scr, &total);
#pragma unroll
            for (int j = 0; j < MSM_P_PER_MAX; j++)
                if (j < PER) { start[t * PER + j] = run; run += c[j]; cnt[t * PER + j] = 0; }
        }
        __syncthreads();
        if (live) {
            if constexpr (C != 0) {
#pragma unroll
                for (int w = 0; w < msm_ct_windows(C); w++) {
                    const uint32_t mag = d[w] & ~MSM_SIGN;
                    if (mag) {
                        const uint32_t key = msm_key(sh, (uint32_t)w, mag, i);
                        const uint32_t p = key >> sh.LB;
                        const uint32_t slot = start[p] + atomicAdd(&cnt[p], 1u);
                        stage[slot] = make_uint2(key, ((uint32_t)((size_t)w * sh.stride) + (uint32_t)i) | (d[w] & MSM_SIGN));
                    }
                }
            } else {
                uint32_t carry = 0;
                uint32_t r[8];
#pragma unroll
                for (int k = 0; k < 8; k++) r[k] = sl[k];
                for (int w = 0; w < sh.W; w++) {
                    uint32_t dg = msm_digit_next(r, sh.c, carry);
                    uint32_t mag = dg & ~MSM_SIGN;
                    if (mag) {
                        uint32_t key = msm_key(sh, (uint32_t)w, mag, i);
                        uint32_t p = key >> sh.LB;
                        uint32_t slot = start[p] + atomicAdd(&cnt[p], 1u);
                        stage[slot] = make_uint2(key, ((uint32_t)((size_t)w * sh.stride) + (uint32_t)i) | (dg & MSM_SIGN));
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t j = t; j < total; j += MSM_SORT_BLOCK) {
            const uint2 e = stage[j];
            const uint32_t p = e.x >> sh.LB;
            const uint32_t dst = goff[p] + (j - start[p]);
            inter_e[dst] = e.y;
            inter_k[dst] = (uint8_t)(e.x & low_mask);
        }
        __syncthreads();
        // (the placement cursors have counted the tile's entries back up: the counts are read from them rather than kept in registers
        // across the tile - the kernel has to fit 56 registers to be placed beside two resident accumulations, msm_acc_persistent.hip)
        for (int j = 0; j < PER; j++) {
            goff[t * PER + j] += cnt[t * PER + j];
            cnt[t * PER + j] = 0;
        }
        __syncthreads();
    }
}

// ---- pass 2: block p sorts partition p by the low key bits ---------------------------------------
// Emits the final sorted entry list, the bucket sizes and the bucket starts of its 2^LB keys.  A partition of <= cap entries is
// sorted into LDS and leaves as one contiguous copy; a larger one (very skewed scalars, or more entries than the partitions can
// split) falls back to scattering straight to global memory.  Both sweeps walk the partition four entries per lane and load: the
// key plane as one u32 (the counting sweep reads nothing else), the entry plane as one uint4.
__global__ __launch_bounds__(MSM_SORT_BLOCK) void msm_part2_kernel(const uint32_t* __restrict__ inter_e, const uint8_t* __restrict__ inter_k,
                                                                     const uint32_t* __restrict__ part_start, uint32_t* __restrict__ sorted,
                                                                     uint32_t* __restrict__ cnt, uint32_t* __restrict__ bucket_start, MsmShape sh,
                                                                     uint32_t cap) {
    msm_sort_wave_prio(sh.low_prio);
    extern __shared__ uint32_t lds[];  // [2^LB] counters, [32] scan scratch, [cap] staged output
    const int p = blockIdx.x, t = threadIdx.x;
    const uint32_t nbins = 1u << sh.LB;
    uint32_t* h = lds;
    uint32_t* scr = lds + nbins;
    uint32_t* stage = scr + 32;
    for (uint32_t b = t; b < nbins; b += MSM_SORT_BLOCK) h[b] = 0;
    __syncthreads();
    const uint32_t lo = part_start[p], hi = part_start[p + 1];
    const uint32_t lo4 = lo & ~3u;  // the planes are walked from the 4-entry boundary below lo (aligned u32 / uint4 loads)
    const bool staged = hi - lo <= cap;
    constexpr int U = 4;  // independent loads in flight per lane (16 entries): the sweeps are latency bound otherwise
    const uint32_t* key_words = reinterpret_cast<const uint32_t*>(inter_k);
    const uint4* entry_quads = reinterpret_cast<const uint4*>(inter_e);
    for (uint32_t base = lo4; base < hi; base += MSM_SORT_BLOCK * 4 * U) {
        uint32_t kw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = base + (uint32_t)(u * MSM_SORT_BLOCK + t) * 4u;
            kw[u] = e < hi ? key_words[e >> 2] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = base + (uint32_t)(u * MSM_SORT_BLOCK + t) * 4u;
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (e + b >= lo && e + b < hi) atomicAdd(&h[(kw[u] >> (8 * b)) & 0xffu], 1u);
        }
    }
    __syncthreads();
    // exclusive scan over the bins: each thread owns a contiguous run of bins
    const uint32_t per = (nbins + MSM_SORT_BLOCK - 1) / MSM_SORT_BLOCK;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) {
        uint32_t b = t * per + j;
        if (b < nbins) sum += h[b];
    }
    uint32_t total;
    uint32_t run = lo + msm_block_scan(sum, scr, &total);
    for (uint32_t j = 0; j < per; j++) {
        uint32_t b = t * per + j;
        if (b < nbins) {
            uint32_t c = h[b];
            size_t key = ((size_t)p << sh.LB) + b;
            cnt[key] = c;
            bucket_start[key] = run;
            h[b] = run;  // becomes the scatter cursor
            run += c;
        }
    }
    __syncthreads();
    for (uint32_t base = lo4; base < hi; base += MSM_SORT_BLOCK * 4 * U) {
        uint32_t kw[U];
        uint4 ev[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = base + (uint32_t)(u * MSM_SORT_BLOCK + t) * 4u;
            if (e < hi) { kw[u] = key_words[e >> 2]; ev[u] = entry_quads[e >> 2]; }
            else { kw[u] = 0u; ev[u] = make_uint4(0, 0, 0, 0); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t e = base + (uint32_t)(u * MSM_SORT_BLOCK + t) * 4u;
            const uint32_t ent[4] = {ev[u].x, ev[u].y, ev[u].z, ev[u].w};
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (e + b >= lo && e + b < hi) {
                    const uint32_t pos = atomicAdd(&h[(kw[u] >> (8 * b)) & 0xffu], 1u);
                    if (staged) stage[pos - lo] = ent[b];
                    else sorted[pos] = ent[b];
                }
            }
        }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t j = t; j < hi - lo; j += MSM_SORT_BLOCK) sorted[lo + j] = stage[j];
    }
}

template <class SF, int C>
static void msm_launch_sort_c(const MsmShape& sh, const void* d_scalars, int is_mont, const MsmSortBufs& b, hipStream_t s) {
    const size_t chunk = (sh.n + MSM_NB1 - 1) / MSM_NB1;
    const size_t entries = (size_t)sh.W * sh.n;
    uint32_t* inter_e = reinterpret_cast<uint32_t*>(b.inter);
    uint8_t* inter_k = reinterpret_cast<uint8_t*>(inter_e + entries);
    allow_dynamic_lds((const void*)msm_scatter1_kernel<C>, (int)MSM_LDS_BYTES);
    allow_dynamic_lds((const void*)msm_part2_kernel, (int)MSM_LDS_BYTES);
    LURK_REQUIRE(msm_scatter1_lds(sh.P, sh.W, sh.tile) <= MSM_LDS_BYTES, "pass-1 tile does not fit the LDS");
    const uint4* canon = (const uint4*)d_scalars;
    if (is_mont) {
        hipLaunchKernelGGL((msm_hist1_kernel<SF, C, true>), dim3(MSM_NB1), dim3(MSM_SORT_BLOCK), (size_t)sh.P * 4, s, (const uint4*)d_scalars,
                           (uint4*)b.canon, b.block_hist, sh, chunk);
        canon = (const uint4*)b.canon;
    } else {
        hipLaunchKernelGGL((msm_hist1_kernel<SF, C, false>), dim3(MSM_NB1), dim3(MSM_SORT_BLOCK), (size_t)sh.P * 4, s, (const uint4*)d_scalars,
                           (uint4*)nullptr, b.block_hist, sh, chunk);
    }
    hipLaunchKernelGGL(msm_scan1_kernel, dim3(sh.P), dim3(MSM_NB1), 0, s, b.block_hist, b.part_cnt, sh.P);
    hipLaunchKernelGGL(msm_part_start_kernel, dim3(1), dim3(MSM_SORT_BLOCK), 0, s, b.part_cnt, b.part_start, sh.P, b.zero[0], b.zero_n[0], b.zero[1],
                       b.zero_n[1], b.zero[2], b.zero_n[2]);
    hipLaunchKernelGGL((msm_scatter1_kernel<C>), dim3(MSM_NB1), dim3(MSM_SORT_BLOCK), msm_scatter1_lds(sh.P, sh.W, sh.tile), s, canon, b.block_hist,
                       b.part_start, inter_e, inter_k, sh, chunk);
    hipLaunchKernelGGL(msm_part2_kernel, dim3(sh.P), dim3(MSM_SORT_BLOCK), MSM_LDS_BYTES, s, inter_e, inter_k, b.part_start, b.sorted, b.cnt,
                       b.bucket_start, sh, (uint32_t)msm_part2_cap(sh.LB));
}

template <class SF>
void msm_launch_sort(const MsmShape& sh, const void* d_scalars, int is_mont, const MsmSortBufs& b, hipStream_t s) {
    // the widths the library chooses itself get the compile-time recoding; an override in between takes the register walk
    if (sh.c == 20) msm_launch_sort_c<SF, 20>(sh, d_scalars, is_mont, b, s);
    else if (sh.c == 16) msm_launch_sort_c<SF, 16>(sh, d_scalars, is_mont, b, s);
    else msm_launch_sort_c<SF, 0>(sh, d_scalars, is_mont, b, s);
}
template void msm_launch_sort<PallasFp>(const MsmShape&, const void*, int, const MsmSortBufs&, hipStream_t);
template void msm_launch_sort<PallasFq>(const MsmShape&, const void*, int, const MsmSortBufs&, hipStream_t);

}  // namespace lurk

// msm.hip - Pedersen multi-scalar multiplication over the Pasta curves for gfx950.
//
// Replaces pasta-msm's mult_pippenger_{pallas,vesta} as reached from arecibo's
// CommitmentEngine::commit (callers: /root/reference/src/proof/nova.rs:287-293,
// /root/reference/src/proof/supernova.rs:231-244).  See msm_core.cuh for the pipeline; this file
// holds the kernels, the resident-bases context and the C ABI.
//
// Memory plan (n scalars, W = 16 windows, G key spaces, B = 2^15 buckets each):
//   bases / table   64 B x n (x W with the precomputed table)   resident for the ctx lifetime
//   digits, sorted  4 B x W x n each                             streamed once per call
//   block_hist      4 B x (W*K) x B = 32 MiB                     LDS-privatised counting sort
//   partials        128 B x (G*B + W*n/S)                        XYZZ task sums
//   buckets         128 B x G*B
// Algorithmic HBM bytes per call: 96 B per point (32 B scalar + 64 B base) - the kernel is bound by
// the integer VALU (v_mad_u64_u32), not by HBM.
#include <memory>

#include "common.hpp"
#include "msm_core.cuh"

namespace lurk {

constexpr int MSM_K = 16;          // chunks per window in the counting sort (W*K = 256 blocks = 1 per CU)
constexpr int MSM_SORT_BLOCK = 1024;
constexpr int MSM_S = 64;          // sorted entries per accumulation task
constexpr int MSM_SMALL = 16;      // buckets with <= this many task partials are summed by one lane
constexpr int MSM_ACC_BLOCK = 256;

// ---- 1. digits ---------------------------------------------------------------------------
template <class SF>  // scalar field
__global__ __launch_bounds__(256) void msm_digits_kernel(const uint4* __restrict__ scalars, uint32_t* __restrict__ digits, size_t n,
                                                           int is_mont) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 lo = scalars[2 * i], hi = scalars[2 * i + 1];
    Fe<SF> s;
    s.l[0] = lo.x; s.l[1] = lo.y; s.l[2] = lo.z; s.l[3] = lo.w;
    s.l[4] = hi.x; s.l[5] = hi.y; s.l[6] = hi.z; s.l[7] = hi.w;
    if (is_mont) s = fe_from_mont<SF>(s);
    uint32_t d[MSM_W];
    msm_scalar_digits(s.l, d);
#pragma unroll
    for (int w = 0; w < MSM_W; w++) digits[(size_t)w * n + i] = d[w];
}

// ---- 2. counting sort ----------------------------------------------------------------------
// block (k, w): LDS histogram of window w over chunk k
__global__ __launch_bounds__(MSM_SORT_BLOCK) void msm_hist_kernel(const uint32_t* __restrict__ digits, uint32_t* __restrict__ block_hist,
                                                                    size_t n, size_t chunk) {
    extern __shared__ uint32_t lds_hist[];
    const int k = blockIdx.x, w = blockIdx.y;
    for (int b = threadIdx.x; b < MSM_B; b += MSM_SORT_BLOCK) lds_hist[b] = 0;
    __syncthreads();
    size_t lo = (size_t)k * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t* src = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += MSM_SORT_BLOCK) {
        uint32_t b = src[i] & ~MSM_SIGN;
        if (b) atomicAdd(&lds_hist[b - 1], 1u);
    }
    __syncthreads();
    uint32_t* dst = block_hist + (size_t)(w * MSM_K + k) * MSM_B;
    for (int b = threadIdx.x; b < MSM_B; b += MSM_SORT_BLOCK) dst[b] = lds_hist[b];
}

// thread per (window w, bin b): exclusive running count down the K chunk rows of window w.
// block_hist[w*K + k][b] becomes the offset of chunk k inside (window w, bucket b); cntw[w][b] the total.
__global__ __launch_bounds__(256) void msm_colscan_kernel(uint32_t* __restrict__ block_hist, uint32_t* __restrict__ cntw) {
    size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;  // w * B + b
    if (id >= (size_t)MSM_W * MSM_B) return;
    size_t w = id / MSM_B, b = id % MSM_B;
    uint32_t* col = block_hist + (w * MSM_K) * MSM_B + b;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < MSM_K; k++) {
        uint32_t v = col[(size_t)k * MSM_B];
        col[(size_t)k * MSM_B] = run;
        run += v;
    }
    cntw[id] = run;
}

// block g (key space), 1024 threads x 32 bins: bucket sizes, bucket starts (absolute positions in
// `sorted`), task starts, and base_off[w][b] = where window w's entries of bucket b begin.
// G == MSM_W: space g is window g.  G == 1: the single space merges all windows (precomputed table).
__global__ __launch_bounds__(1024) void msm_binscan_kernel(const uint32_t* __restrict__ cntw, int G, uint32_t* __restrict__ cnt,
                                                             uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ task_start,
                                                             uint32_t* __restrict__ space_tasks, uint32_t* __restrict__ base_off,
                                                             size_t space_stride) {
    __shared__ uint32_t sh_a[1024], sh_b[1024];
    const int g = blockIdx.x, t = threadIdx.x;
    constexpr int PER = MSM_B / 1024;  // 32 bins per thread
    const int nwin = G == 1 ? MSM_W : 1, w0 = G == 1 ? 0 : g;
    uint32_t c[PER];
    uint32_t tot = 0, ttot = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        uint32_t v = 0;
        for (int w = 0; w < nwin; w++) v += cntw[(size_t)(w0 + w) * MSM_B + (size_t)t * PER + j];
        c[j] = v;
        tot += v;
        ttot += (v + MSM_S - 1) / MSM_S;
    }
    sh_a[t] = tot;
    sh_b[t] = ttot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        uint32_t a = t >= off ? sh_a[t - off] : 0, b = t >= off ? sh_b[t - off] : 0;
        __syncthreads();
        sh_a[t] += a;
        sh_b[t] += b;
        __syncthreads();
    }
    uint32_t run = (uint32_t)((size_t)g * space_stride) + sh_a[t] - tot, trun = sh_b[t] - ttot;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        size_t bin = (size_t)t * PER + j, gb = (size_t)g * MSM_B + bin;
        cnt[gb] = c[j];
        bucket_start[gb] = run;
        task_start[(size_t)g * (MSM_B + 1) + bin] = trun;
        uint32_t wrun = run;
        for (int w = 0; w < nwin; w++) {
            base_off[(size_t)(w0 + w) * MSM_B + bin] = wrun;
            wrun += cntw[(size_t)(w0 + w) * MSM_B + bin];
        }
        run += c[j];
        trun += (c[j] + MSM_S - 1) / MSM_S;
    }
    if (t == 1023) {
        task_start[(size_t)g * (MSM_B + 1) + MSM_B] = trun;
        space_tasks[g] = trun;
    }
}

// exclusive scan of the per-space task totals (G <= 16) -> space_task_base[0..G]
__global__ void msm_task_base_kernel(const uint32_t* __restrict__ space_tasks, uint32_t* __restrict__ space_task_base, int G) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (int g = 0; g < G; g++) {
            space_task_base[g] = run;
            run += space_tasks[g];
        }
        space_task_base[G] = run;
    }
}

// block (k, w): scatter entries of window w / chunk k to their sorted positions
__global__ __launch_bounds__(MSM_SORT_BLOCK) void msm_scatter_kernel(const uint32_t* __restrict__ digits,
                                                                       const uint32_t* __restrict__ block_off,
                                                                       const uint32_t* __restrict__ base_off, uint32_t* __restrict__ sorted,
                                                                       size_t n, size_t chunk, size_t table_stride_per_window) {
    extern __shared__ uint32_t lds_off[];
    const int k = blockIdx.x, w = blockIdx.y;
    const uint32_t* off = block_off + (size_t)(w * MSM_K + k) * MSM_B;
    const uint32_t* boff = base_off + (size_t)w * MSM_B;
    for (int b = threadIdx.x; b < MSM_B; b += MSM_SORT_BLOCK) lds_off[b] = off[b] + boff[b];
    __syncthreads();
    size_t lo = (size_t)k * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t* src = digits + (size_t)w * n;
    const uint32_t tbase = (uint32_t)((size_t)w * table_stride_per_window);
    for (size_t i = lo + threadIdx.x; i < hi; i += MSM_SORT_BLOCK) {
        uint32_t d = src[i];
        uint32_t b = d & ~MSM_SIGN;
        if (b) {
            uint32_t pos = atomicAdd(&lds_off[b - 1], 1u);
            sorted[pos] = (tbase + (uint32_t)i) | (d & MSM_SIGN);
        }
    }
}

// ---- 3. accumulate -------------------------------------------------------------------------
// task table: task t -> [first, last) of the sorted list (<= MSM_S entries of one bucket)
__global__ __launch_bounds__(256) void msm_tasks_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bucket_start,
                                                          const uint32_t* __restrict__ task_start,
                                                          const uint32_t* __restrict__ space_task_base, int G, uint2* __restrict__ task_info) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= space_task_base[G]) return;
    int g = 0;
    while (g + 1 < G && space_task_base[g + 1] <= t) g++;
    uint32_t tl = t - space_task_base[g];
    const uint32_t* ts = task_start + (size_t)g * (MSM_B + 1);
    uint32_t b = msm_upper_slot(ts, MSM_B, tl);
    uint32_t part = tl - ts[b];
    size_t gb = (size_t)g * MSM_B + b;
    uint32_t first = bucket_start[gb] + part * MSM_S;
    uint32_t end = bucket_start[gb] + cnt[gb];
    task_info[t] = make_uint2(first, first + MSM_S < end ? first + MSM_S : end);
}

// Longest-task-first order: tasks are counting-sorted by length (1..MSM_S) in descending order, so
// the 64 lanes of a wave run tasks of equal length (no lane waits for the longest task of its wave;
// bucket sizes are ragged - Poisson around n/2^15 per window - and skewed for witness-like scalars)
// and the short tasks fill the tail of the launch.
__global__ __launch_bounds__(256) void msm_len_hist_kernel(const uint2* __restrict__ task_info, const uint32_t* __restrict__ space_task_base,
                                                             int G, uint32_t* __restrict__ len_hist) {
    __shared__ uint32_t sh[MSM_S + 1];
    if (threadIdx.x <= MSM_S) sh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < space_task_base[G]) {
        uint2 ti = task_info[t];
        atomicAdd(&sh[ti.y - ti.x], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= MSM_S && sh[threadIdx.x]) atomicAdd(&len_hist[threadIdx.x], sh[threadIdx.x]);
}
// len_hist -> start offset of each length class, longest first (single small block)
__global__ void msm_len_scan_kernel(uint32_t* __restrict__ len_hist, uint32_t* __restrict__ len_cursor) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (int l = MSM_S; l >= 0; l--) {
            len_cursor[l] = run;
            run += len_hist[l];
        }
    }
}
__global__ __launch_bounds__(256) void msm_len_scatter_kernel(const uint2* __restrict__ task_info, const uint32_t* __restrict__ space_task_base,
                                                                int G, uint32_t* __restrict__ len_cursor, uint32_t* __restrict__ order) {
    __shared__ uint32_t sh_cnt[MSM_S + 1], sh_base[MSM_S + 1];
    if (threadIdx.x <= MSM_S) sh_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    bool valid = t < space_task_base[G];
    uint32_t len = 0, rank = 0;
    if (valid) {
        uint2 ti = task_info[t];
        len = ti.y - ti.x;
        rank = atomicAdd(&sh_cnt[len], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= MSM_S && sh_cnt[threadIdx.x]) sh_base[threadIdx.x] = atomicAdd(&len_cursor[threadIdx.x], sh_cnt[threadIdx.x]);
    __syncthreads();
    if (valid) order[sh_base[len] + rank] = t;
}

template <class P>
__global__ __launch_bounds__(MSM_ACC_BLOCK) void msm_accumulate_kernel(const uint32_t* __restrict__ sorted, const Affine<P>* __restrict__ table,
                                                                         const uint2* __restrict__ task_info, const uint32_t* __restrict__ order,
                                                                         const uint32_t* __restrict__ space_task_base, int G,
                                                                         Xyzz<P>* __restrict__ partials) {
    uint32_t i = blockIdx.x * MSM_ACC_BLOCK + threadIdx.x;
    if (i >= space_task_base[G]) return;
    uint32_t t = order[i];
    uint2 ti = task_info[t];
    partials[t] = msm_task_accumulate<P>(sorted, ti.x, ti.y, table);
}

// ---- 4. finalize ---------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void msm_finalize_kernel(const Xyzz<P>* __restrict__ partials, const uint32_t* __restrict__ cnt,
                                                             const uint32_t* __restrict__ task_start,
                                                             const uint32_t* __restrict__ space_task_base, int G, Xyzz<P>* __restrict__ buckets,
                                                             uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count) {
    size_t gb = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gb >= (size_t)G * MSM_B) return;
    int g = (int)(gb / MSM_B);
    uint32_t b = (uint32_t)(gb % MSM_B);
    uint32_t nt = (cnt[gb] + MSM_S - 1) / MSM_S;
    uint32_t first = space_task_base[g] + task_start[(size_t)g * (MSM_B + 1) + b];
    if (nt > MSM_SMALL) {
        big_list[atomicAdd(big_count, 1u)] = (uint32_t)gb;
        return;
    }
    Xyzz<P> acc = xyzz_identity<P>();
    for (uint32_t i = 0; i < nt; i++) xyzz_add<P>(acc, partials[first + i]);
    buckets[gb] = acc;
}

template <class P, int BLOCK>
__device__ void block_tree_sum(Xyzz<P>& acc, Xyzz<P>* sh) {
    const int t = threadIdx.x;
    sh[t] = acc;
    __syncthreads();
    for (int stride = BLOCK / 2; stride >= 1; stride >>= 1) {
        if (t < stride) {
            xyzz_add<P>(acc, sh[t + stride]);
            sh[t] = acc;
        }
        __syncthreads();
    }
}

// hot buckets (more than MSM_SMALL partials): one workgroup each, lanes stride the partials
template <class P>
__global__ __launch_bounds__(256) void msm_big_bucket_kernel(const Xyzz<P>* __restrict__ partials, const uint32_t* __restrict__ cnt,
                                                               const uint32_t* __restrict__ task_start,
                                                               const uint32_t* __restrict__ space_task_base, Xyzz<P>* __restrict__ buckets,
                                                               const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count) {
    extern __shared__ uint4 lds_raw[];
    Xyzz<P>* sh = reinterpret_cast<Xyzz<P>*>(lds_raw);
    const uint32_t nbig = *big_count;
    for (uint32_t i = blockIdx.x; i < nbig; i += gridDim.x) {
        uint32_t gb = big_list[i];
        int g = gb / MSM_B;
        uint32_t b = gb % MSM_B;
        uint32_t nt = (cnt[gb] + MSM_S - 1) / MSM_S;
        uint32_t first = space_task_base[g] + task_start[(size_t)g * (MSM_B + 1) + b];
        Xyzz<P> acc = xyzz_identity<P>();
        for (uint32_t j = threadIdx.x; j < nt; j += 256) xyzz_add<P>(acc, partials[first + j]);
        block_tree_sum<P, 256>(acc, sh);
        if (threadIdx.x == 0) buckets[gb] = acc;
        __syncthreads();
    }
}

// ---- 5. bucket reduction -------------------------------------------------------------------
// sum_b b*B_b with the bucket b stored at index idx = b-1:  sum (idx+1) X_idx = S + sum_k 2^k P_k,
// S = sum X, P_k = sum of the X whose idx has bit k set.  The (S, P_0..P_{k-1}) vectors of two
// adjacent segments of 2^k items merge with k+1 independent additions (the new plane k is the upper
// half's S), so the whole reduction is 15 levels of depth ONE addition each: these tail kernels are
// latency bound (~13 us per dependent XYZZ addition) and a running-sum formulation needs >100 of them.
// level k: in[g][seg][0..k] (segments of 2^k items) -> out[g][seg/2][0..k+1]
template <class P>
__global__ __launch_bounds__(256) void msm_planes_kernel(const Xyzz<P>* __restrict__ in, Xyzz<P>* __restrict__ out, int k, int G) {
    const size_t nseg_out = (size_t)MSM_B >> (k + 1);
    const size_t comps_out = (size_t)k + 2, comps_in = (size_t)k + 1;
    size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)G * nseg_out * comps_out) return;
    size_t c = id % comps_out, seg = (id / comps_out) % nseg_out, g = id / (comps_out * nseg_out);
    const Xyzz<P>* lo = in + ((g * (nseg_out * 2) + 2 * seg) * comps_in);
    const Xyzz<P>* hi = lo + comps_in;
    Xyzz<P> r;
    if (c <= (size_t)k) {
        r = lo[c];
        xyzz_add<P>(r, hi[c]);
    } else {
        r = hi[0];
    }
    out[id] = r;
}
// block g, 16 lanes: W = S + sum_k 2^k P_k by a tree-shaped Horner (15 doublings + 5 additions deep)
template <class P>
__global__ __launch_bounds__(64) void msm_horner_kernel(const Xyzz<P>* __restrict__ planes, Xyzz<P>* __restrict__ ws) {
    __shared__ uint4 lds_raw[16 * sizeof(Xyzz<P>) / 16];
    Xyzz<P>* sh = reinterpret_cast<Xyzz<P>*>(lds_raw);
    const int g = blockIdx.x, t = threadIdx.x;
    const Xyzz<P>* v = planes + (size_t)g * (MSM_C);  // [S, P_0 .. P_14]
    if (t < 16) sh[t] = t < MSM_C - 1 ? v[1 + t] : xyzz_identity<P>();
    __syncthreads();
    for (int lvl = 0; lvl < 4; lvl++) {  // q_j = q_2j + 2^(2^lvl) * q_2j+1
        Xyzz<P> r;
        bool active = t < (8 >> lvl);
        if (active) {
            r = xyzz_dbl_n<P>(sh[2 * t + 1], 1 << lvl);
            xyzz_add<P>(r, sh[2 * t]);
        }
        __syncthreads();
        if (active) sh[t] = r;
        __syncthreads();
    }
    if (t == 0) {
        Xyzz<P> r = sh[0];
        xyzz_add<P>(r, v[0]);
        ws[g] = r;
    }
}

// ---- precomputed table: T[w*n + i] = 2^(16 w) * P_i ------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void msm_precompute_kernel(const Affine<P>* __restrict__ bases, size_t n, Affine<P>* __restrict__ table) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Affine<P> a = bases[i];
    table[i] = a;
    Xyzz<P> p = xyzz_from_affine<P>(a);
    for (int w = 1; w < MSM_W; w++) {
        p = xyzz_dbl_n<P>(p, MSM_C);
        Affine<P> q = xyzz_to_affine<P>(p);
        table[(size_t)w * n + i] = q;
        p = xyzz_from_affine<P>(q);  // keep ZZ = 1: cheaper doublings, smaller drift
    }
}

// ---- context -------------------------------------------------------------------------------
struct MsmCtxBase {
    int curve = 0;
    size_t npoints = 0;
    bool precomputed = false;
    virtual ~MsmCtxBase() {}
    virtual void run(const void* d_scalars, size_t n, int is_mont, hipStream_t s, void* out_jac96_host) = 0;
};

template <class P, class SF>
struct MsmCtx : MsmCtxBase {
    DevBuf own_bases;                // bases (or the whole table when precomputed)
    const Affine<P>* table = nullptr;
    size_t table_stride = 0;         // = npoints when precomputed (window w at w*npoints), else 0
    std::mutex mu;
    // workspace
    DevBuf digits, sorted, block_hist, cntw, base_off, cnt, bucket_start, task_start, space_tasks, space_task_base, task_info, task_order, len_hist, partials,
        buckets, big_list, big_count, planes_a, planes_b, ws;
    Xyzz<P>* ws_host = nullptr;
    size_t ws_n = 0;

    ~MsmCtx() override {
        if (ws_host) (void)hipHostFree(ws_host);
    }
    int G() const { return precomputed ? 1 : MSM_W; }

    void set_bases_device(const void* d_bases, size_t n, bool copy, bool precompute, hipStream_t s) {
        npoints = n;
        precomputed = precompute;
        LURK_REQUIRE(n < ((size_t)1 << 31) / (precompute ? MSM_W : 1), "too many points for 31-bit table indices");
        if (precompute) {
            own_bases.alloc((size_t)MSM_W * n * sizeof(Affine<P>));
            if (n) {
                ProfScope ps("msm_precompute", s);
                hipLaunchKernelGGL((msm_precompute_kernel<P>), dim3(div_up(n, 256)), dim3(256), 0, s, (const Affine<P>*)d_bases, n,
                                   own_bases.as<Affine<P>>());
                LURK_HIP_CHECK(hipGetLastError());
            }
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            table = own_bases.as<Affine<P>>();
            table_stride = n;
        } else if (copy) {
            own_bases.alloc(n * sizeof(Affine<P>));
            LURK_HIP_CHECK(hipMemcpyAsync(own_bases.p, d_bases, n * sizeof(Affine<P>), hipMemcpyDeviceToDevice, s));
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            table = own_bases.as<Affine<P>>();
        } else {
            table = (const Affine<P>*)d_bases;  // borrowed
        }
    }

    void ensure_workspace(size_t n) {
        if (n <= ws_n && ws_n != 0) return;
        const int g = G();
        size_t ntask_max = (size_t)g * MSM_B + (size_t)MSM_W * n / MSM_S + MSM_W + 1;
        digits.ensure((size_t)MSM_W * n * 4);
        sorted.ensure((size_t)MSM_W * n * 4);
        block_hist.ensure((size_t)MSM_W * MSM_K * MSM_B * 4);
        cnt.ensure((size_t)g * MSM_B * 4);
        bucket_start.ensure((size_t)g * MSM_B * 4);
        task_start.ensure((size_t)g * (MSM_B + 1) * 4);
        space_tasks.ensure(64 * 4);
        space_task_base.ensure(64 * 4);
        partials.ensure(ntask_max * sizeof(Xyzz<P>));
        task_info.ensure(ntask_max * sizeof(uint2));
        task_order.ensure(ntask_max * 4);
        len_hist.ensure(2 * (MSM_S + 1) * 4);
        buckets.ensure((size_t)g * MSM_B * sizeof(Xyzz<P>));
        big_list.ensure((size_t)g * MSM_B * 4);
        big_count.ensure(16);
        cntw.ensure((size_t)MSM_W * MSM_B * 4);
        base_off.ensure((size_t)MSM_W * MSM_B * 4);
        planes_a.ensure((size_t)g * MSM_B * sizeof(Xyzz<P>));  // level k holds (B >> (k+1)) * (k+2) <= B points per space
        planes_b.ensure((size_t)g * MSM_B * sizeof(Xyzz<P>));
        ws.ensure((size_t)MSM_W * sizeof(Xyzz<P>));
        if (!ws_host) LURK_HIP_CHECK(hipHostMalloc((void**)&ws_host, MSM_W * sizeof(Xyzz<P>)));
        ws_n = n;
    }

    void run(const void* d_scalars, size_t n, int is_mont, hipStream_t s, void* out_jac96_host) override {
        std::lock_guard<std::mutex> lk(mu);
        LURK_REQUIRE(n <= npoints, "more scalars than bases in the context");
        Jacobian<P>* out = (Jacobian<P>*)out_jac96_host;
        if (n == 0) {
            *out = jacobian_from_affine<P>(Affine<P>{fe_zero<P>(), fe_zero<P>()});
            return;
        }
        ensure_workspace(n);
        const int g = G();
        const size_t chunk = (n + MSM_K - 1) / MSM_K;
        const size_t lds = (size_t)MSM_B * 4;
        {
            ProfScope ps("msm_digits", s);
            hipLaunchKernelGGL((msm_digits_kernel<SF>), dim3(div_up(n, 256)), dim3(256), 0, s, (const uint4*)d_scalars, digits.as<uint32_t>(), n,
                               is_mont);
        }
        {
            static bool attr = false;
            if (!attr) {
                LURK_HIP_CHECK(hipFuncSetAttribute((const void*)msm_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                LURK_HIP_CHECK(hipFuncSetAttribute((const void*)msm_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr = true;
            }
            ProfScope ps("msm_sort", s);
            hipLaunchKernelGGL(msm_hist_kernel, dim3(MSM_K, MSM_W), dim3(MSM_SORT_BLOCK), lds, s, digits.as<uint32_t>(),
                               block_hist.as<uint32_t>(), n, chunk);
            // generic: space g = window g owns sorted[g*n, (g+1)*n); precomputed: one space owning sorted[0, W*n)
            hipLaunchKernelGGL(msm_colscan_kernel, dim3(div_up((size_t)MSM_W * MSM_B, 256)), dim3(256), 0, s, block_hist.as<uint32_t>(),
                               cntw.as<uint32_t>());
            hipLaunchKernelGGL(msm_binscan_kernel, dim3(g), dim3(1024), 0, s, cntw.as<uint32_t>(), g, cnt.as<uint32_t>(),
                               bucket_start.as<uint32_t>(), task_start.as<uint32_t>(), space_tasks.as<uint32_t>(), base_off.as<uint32_t>(),
                               precomputed ? (size_t)0 : n);
            hipLaunchKernelGGL(msm_task_base_kernel, dim3(1), dim3(64), 0, s, space_tasks.as<uint32_t>(), space_task_base.as<uint32_t>(), g);
            hipLaunchKernelGGL(msm_scatter_kernel, dim3(MSM_K, MSM_W), dim3(MSM_SORT_BLOCK), lds, s, digits.as<uint32_t>(),
                               block_hist.as<uint32_t>(), base_off.as<uint32_t>(), sorted.as<uint32_t>(), n, chunk, table_stride);
        }
        const size_t ntask_max = (size_t)g * MSM_B + (size_t)MSM_W * n / MSM_S + MSM_W + 1;
        {
            ProfScope ps("msm_tasks", s);
            LURK_HIP_CHECK(hipMemsetAsync(big_count.p, 0, 4, s));
            LURK_HIP_CHECK(hipMemsetAsync(len_hist.p, 0, 2 * (MSM_S + 1) * 4, s));
            uint32_t* lh = len_hist.as<uint32_t>();
            hipLaunchKernelGGL(msm_tasks_kernel, dim3(div_up(ntask_max, 256)), dim3(256), 0, s, cnt.as<uint32_t>(), bucket_start.as<uint32_t>(),
                               task_start.as<uint32_t>(), space_task_base.as<uint32_t>(), g, task_info.as<uint2>());
            hipLaunchKernelGGL(msm_len_hist_kernel, dim3(div_up(ntask_max, 256)), dim3(256), 0, s, task_info.as<uint2>(),
                               space_task_base.as<uint32_t>(), g, lh);
            hipLaunchKernelGGL(msm_len_scan_kernel, dim3(1), dim3(64), 0, s, lh, lh + MSM_S + 1);
            hipLaunchKernelGGL(msm_len_scatter_kernel, dim3(div_up(ntask_max, 256)), dim3(256), 0, s, task_info.as<uint2>(),
                               space_task_base.as<uint32_t>(), g, lh + MSM_S + 1, task_order.as<uint32_t>());
        }
        {
            ProfScope ps("msm_accumulate", s);
            hipLaunchKernelGGL((msm_accumulate_kernel<P>), dim3(div_up(ntask_max, MSM_ACC_BLOCK)), dim3(MSM_ACC_BLOCK), 0, s,
                               sorted.as<uint32_t>(), table, task_info.as<uint2>(), task_order.as<uint32_t>(), space_task_base.as<uint32_t>(), g,
                               partials.as<Xyzz<P>>());
        }
        {
            ProfScope ps("msm_finalize", s);
            hipLaunchKernelGGL((msm_finalize_kernel<P>), dim3(div_up((size_t)g * MSM_B, 256)), dim3(256), 0, s, partials.as<Xyzz<P>>(),
                               cnt.as<uint32_t>(), task_start.as<uint32_t>(), space_task_base.as<uint32_t>(), g, buckets.as<Xyzz<P>>(),
                               big_list.as<uint32_t>(), big_count.as<uint32_t>());
            hipLaunchKernelGGL((msm_big_bucket_kernel<P>), dim3(512), dim3(256), 256 * sizeof(Xyzz<P>), s, partials.as<Xyzz<P>>(),
                               cnt.as<uint32_t>(), task_start.as<uint32_t>(), space_task_base.as<uint32_t>(), buckets.as<Xyzz<P>>(),
                               big_list.as<uint32_t>(), big_count.as<uint32_t>());
        }
        {
            ProfScope ps("msm_reduce", s);
            const Xyzz<P>* in = buckets.as<Xyzz<P>>();
            Xyzz<P>* bufs[2] = {planes_a.as<Xyzz<P>>(), planes_b.as<Xyzz<P>>()};
            for (int k = 0; k < MSM_C - 1; k++) {
                size_t threads = (size_t)g * ((size_t)MSM_B >> (k + 1)) * (k + 2);
                hipLaunchKernelGGL((msm_planes_kernel<P>), dim3(div_up(threads, 256)), dim3(256), 0, s, in, bufs[k & 1], k, g);
                in = bufs[k & 1];
            }
            hipLaunchKernelGGL((msm_horner_kernel<P>), dim3(g), dim3(64), 0, s, in, ws.as<Xyzz<P>>());
        }
        LURK_HIP_CHECK(hipGetLastError());
        LURK_HIP_CHECK(hipMemcpyAsync(ws_host, ws.p, (size_t)g * sizeof(Xyzz<P>), hipMemcpyDeviceToHost, s));
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        // host tail over <= 16 points: sum_w 2^(16w) * W_w (240 sequential doublings)
        Xyzz<P> total = msm_combine_windows<P>(ws_host, g);
        *out = jacobian_from_affine<P>(xyzz_to_affine<P>(total));
    }
};

static MsmCtxBase* new_ctx(int curve) {
    LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
    MsmCtxBase* c = curve == LURK_CURVE_PALLAS ? (MsmCtxBase*)new MsmCtx<PallasFp, PallasFq>() : (MsmCtxBase*)new MsmCtx<PallasFq, PallasFp>();
    c->curve = curve;
    return c;
}
static void ctx_set_bases(MsmCtxBase* c, const void* d_bases, size_t n, bool copy, bool pre, hipStream_t s) {
    if (c->curve == LURK_CURVE_PALLAS) static_cast<MsmCtx<PallasFp, PallasFq>*>(c)->set_bases_device(d_bases, n, copy, pre, s);
    else static_cast<MsmCtx<PallasFq, PallasFp>*>(c)->set_bases_device(d_bases, n, copy, pre, s);
}

template <class P>
static void point_sum_host(const void* pts, size_t count, void* out) {
    const Jacobian<P>* in = (const Jacobian<P>*)pts;
    Xyzz<P> acc = xyzz_identity<P>();
    for (size_t i = 0; i < count; i++) xyzz_add<P>(acc, xyzz_from_jacobian<P>(in[i]));
    *(Jacobian<P>*)out = jacobian_from_affine<P>(xyzz_to_affine<P>(acc));
}
template <class P>
static void point_affine_canonical_host(const void* pt, void* out) {
    Affine<P> a = xyzz_to_affine<P>(xyzz_from_jacobian<P>(*(const Jacobian<P>*)pt));
    Fe<P> x = fe_from_mont<P>(a.x), y = fe_from_mont<P>(a.y);
    memcpy(out, x.l, 32);
    memcpy((char*)out + 32, y.l, 32);
}

static int msm_oneshot(int curve, void* out, const void* bases, size_t n, const void* scalars, int is_mont) {
    return guarded([&] {
        LURK_REQUIRE(out, "null output");
        LURK_REQUIRE(n == 0 || (bases && scalars), "null buffer");
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        DevBuf db(n * 64), ds(n * 32);
        if (n) {
            LURK_HIP_CHECK(hipMemcpy(db.p, bases, n * 64, hipMemcpyHostToDevice));
            LURK_HIP_CHECK(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
        }
        ctx_set_bases(c.get(), db.p, n, false, false, nullptr);
        c->run(ds.p, n, is_mont, nullptr, out);
    });
}

}  // namespace lurk

using namespace lurk;

struct lurk_hip_msm_ctx {
    std::unique_ptr<MsmCtxBase> impl;
};

extern "C" {

int lurk_hip_msm_pallas(void* out, const void* bases, size_t n, const void* scalars, int is_mont) {
    return msm_oneshot(LURK_CURVE_PALLAS, out, bases, n, scalars, is_mont);
}
int lurk_hip_msm_vesta(void* out, const void* bases, size_t n, const void* scalars, int is_mont) {
    return msm_oneshot(LURK_CURVE_VESTA, out, bases, n, scalars, is_mont);
}

int lurk_hip_msm_ctx_create(lurk_hip_msm_ctx** ctx, int curve, const void* bases, size_t n, int flags) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx pointer");
        LURK_REQUIRE(n == 0 || bases, "null bases");
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        bool pre = (flags & LURK_MSM_FLAG_PRECOMPUTE) != 0;
        DevBuf tmp(n * 64);
        if (n) LURK_HIP_CHECK(hipMemcpy(tmp.p, bases, n * 64, hipMemcpyHostToDevice));
        ctx_set_bases(c.get(), tmp.p, n, /*copy=*/true, pre, nullptr);
        *ctx = new lurk_hip_msm_ctx{std::move(c)};
    });
}
int lurk_hip_msm_ctx_create_dev(lurk_hip_msm_ctx** ctx, int curve, const void* d_bases, size_t n, int flags, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx pointer");
        LURK_REQUIRE(n == 0 || d_bases, "null bases");
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        bool pre = (flags & LURK_MSM_FLAG_PRECOMPUTE) != 0;
        ctx_set_bases(c.get(), d_bases, n, /*copy=*/false, pre, (hipStream_t)stream);
        *ctx = new lurk_hip_msm_ctx{std::move(c)};
    });
}
int lurk_hip_msm_ctx_run(lurk_hip_msm_ctx* ctx, void* out, const void* scalars, size_t n, int is_mont) {
    return guarded([&] {
        LURK_REQUIRE(ctx && out, "null argument");
        LURK_REQUIRE(n == 0 || scalars, "null scalars");
        DevBuf ds(n * 32);
        if (n) LURK_HIP_CHECK(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
        ctx->impl->run(ds.p, n, is_mont, nullptr, out);
    });
}
int lurk_hip_msm_ctx_run_dev(lurk_hip_msm_ctx* ctx, void* out, const void* d_scalars, size_t n, int is_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(ctx && out, "null argument");
        LURK_REQUIRE(n == 0 || d_scalars, "null scalars");
        ctx->impl->run(d_scalars, n, is_mont, (hipStream_t)stream, out);
    });
}
int lurk_hip_msm_ctx_destroy(lurk_hip_msm_ctx* ctx) {
    delete ctx;
    return 0;
}

// host-side group helpers (a handful of points: partial commitments gathered from the ranks)
int lurk_hip_point_sum(int curve, void* out, const void* points, size_t count) {
    try {
        LURK_REQUIRE(curve == 0 || curve == 1, "unknown curve id");
        LURK_REQUIRE(out && (count == 0 || points), "null argument");
        if (curve == 0) point_sum_host<PallasFp>(points, count, out);
        else point_sum_host<PallasFq>(points, count, out);
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}
int lurk_hip_point_to_affine_canonical(int curve, void* out_xy64, const void* point) {
    try {
        LURK_REQUIRE(curve == 0 || curve == 1, "unknown curve id");
        LURK_REQUIRE(out_xy64 && point, "null argument");
        if (curve == 0) point_affine_canonical_host<PallasFp>(point, out_xy64);
        else point_affine_canonical_host<PallasFq>(point, out_xy64);
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}
}

// msm.hip - Pedersen multi-scalar multiplication over the Pasta curves for gfx950.
//
// Replaces pasta-msm's mult_pippenger_{pallas,vesta} as reached from arecibo's
// CommitmentEngine::commit (callers: /root/reference/src/proof/nova.rs:287-293,
// /root/reference/src/proof/supernova.rs:231-244).  See msm_core.cuh for the pipeline; this file
// holds the kernels, the resident-bases context and the C ABI.
//
// Memory plan (n scalars, c-bit windows, W windows, NB = G * 2^(c-1) keys <= 2^19):
//   bases / table   64 B x n (x W with the precomputed table)   resident for the ctx lifetime
//   scalars         32 B x n, read by both sweeps of sort pass 1  caller's
//   inter           8 B x W x n  (low key bits, entry)           between the two sort passes
//   sorted          4 B x W x n  (table index | sign)
//   partials        128 B x (NB + W*n/S)                         XYZZ task sums
//   buckets, planes 128 B x NB (x3)
// Algorithmic HBM bytes per call: 96 B per point (32 B scalar + 64 B base) - the dominant kernel is
// bound by the integer VALU (v_mad_u64_u32 issue), not by HBM.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <memory>

#include "common.hpp"
#include "msm_core.cuh"
#include "msm_sort.hpp"

namespace lurk {

void keygen_from_label_device(int curve, const void* label, size_t label_len, size_t n, void* d_out, hipStream_t s);  // keygen.hip

constexpr int MSM_SMALL = 16;      // buckets with <= this many task partials are summed by one lane (msm_finalize.hip: MSM_FIN_SMALL)
constexpr int MSM_ACC_BLOCK = 256;

// Every kernel of a commitment except the bucket accumulation is short and bound by latency, LDS atomics or HBM; with
// commitments in flight they share the SIMDs with the (older, VALU-saturating) accumulate waves of the previous
// commitment, and the instruction arbiter serves the oldest wave first: measured 15-18x slowdowns of these kernels.
// Raising their wave priority lets them issue when they are ready; they need a few percent of the VALU.
// `low`: the commitment was submitted with LURK_MSM_SUBMIT_FOLLOW - work staged ahead that must only take what the open step's serial
// chain (cross term, commit(T), folds: wave priority 3) leaves; its sort and plan kernels then run at the lowest wave priority like its
// accumulation (the tail kernels - finalize, bucket reduction - always run at 3: see submit_impl).
__device__ __forceinline__ void msm_set_wave_prio(int low) {
    if (low) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(3);
}

// ---- 1-2. digits and sort: msm_sort.hip (msm_launch_sort) ---------------------------------------------

// ---- 3. task planning ------------------------------------------------------------------------
// block g (group of MSM_GRP keys), 1024 threads x 32 keys: task starts inside the group + group total
__global__ __launch_bounds__(1024) void msm_taskscan_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ task_start,
                                                              uint32_t* __restrict__ group_tasks, uint32_t S, int low) {
    msm_set_wave_prio(low);
    __shared__ uint32_t sh[1024];
    const int g = blockIdx.x, t = threadIdx.x;
    constexpr int PER = MSM_GRP / 1024;
    const uint4* src = reinterpret_cast<const uint4*>(cnt + (size_t)g * MSM_GRP + (size_t)t * PER);
    uint32_t c[PER];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < PER / 4; j++) {
        uint4 v = src[j];
        c[4 * j] = (v.x + S - 1) / S;
        c[4 * j + 1] = (v.y + S - 1) / S;
        c[4 * j + 2] = (v.z + S - 1) / S;
        c[4 * j + 3] = (v.w + S - 1) / S;
        tot += c[4 * j] + c[4 * j + 1] + c[4 * j + 2] + c[4 * j + 3];
    }
    sh[t] = tot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t a = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += a;
        __syncthreads();
    }
    uint32_t run = sh[t] - tot;
    uint32_t* dst = task_start + (size_t)g * (MSM_GRP + 1) + (size_t)t * PER;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        dst[j] = run;
        run += c[j];
    }
    if (t == 1023) {
        task_start[(size_t)g * (MSM_GRP + 1) + MSM_GRP] = run;
        group_tasks[g] = run;
    }
}
// task table: task t -> [first, last) of the sorted list (<= S entries of one bucket).  Every workgroup first scans the per-group task
// totals itself (NG <= 32 values: group_task_base[0..NG], which workgroup 0 also stores for the kernels that follow) - a launch of its
// own for that scan was one more link in a chain of dependent launches that costs 5-60 us per link beside resident accumulations.
constexpr int MSM_NG_MAX = 64;
__global__ __launch_bounds__(256) void msm_tasks_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bucket_start,
                                                          const uint32_t* __restrict__ task_start, const uint32_t* __restrict__ group_tasks,
                                                          uint32_t* __restrict__ group_task_base_out, int NG, uint2* __restrict__ task_info, uint32_t S, int low) {
    msm_set_wave_prio(low);
    __shared__ uint32_t group_task_base[MSM_NG_MAX + 1];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int g = 0; g < NG; g++) {
            group_task_base[g] = run;
            run += group_tasks[g];
        }
        group_task_base[NG] = run;
        if (blockIdx.x == 0)
            for (int g = 0; g <= NG; g++) group_task_base_out[g] = group_task_base[g];
    }
    __syncthreads();
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= group_task_base[NG]) return;
    int g = 0;
    while (g + 1 < NG && group_task_base[g + 1] <= t) g++;
    uint32_t tl = t - group_task_base[g];
    const uint32_t* ts = task_start + (size_t)g * (MSM_GRP + 1);
    uint32_t b = msm_upper_slot(ts, MSM_GRP, tl);
    uint32_t part = tl - ts[b];
    size_t key = (size_t)g * MSM_GRP + b;
    uint32_t first = bucket_start[key] + part * S;
    uint32_t end = bucket_start[key] + cnt[key];
    task_info[t] = make_uint2(first, first + S < end ? first + S : end);
}

// Longest-task-first order: tasks are counting-sorted by length (1..MSM_S) in descending order, so
// the 64 lanes of a wave run tasks of equal length (no lane waits for the longest task of its wave;
// bucket sizes are ragged - Poisson around their mean - and skewed for witness-like scalars) and the
// short tasks fill the tail of the launch.  Full tasks (length MSM_S, the bulk at large n) are
// counted per wave with one ballot instead of one LDS atomic each.
__global__ __launch_bounds__(1024) void msm_len_hist_kernel(const uint2* __restrict__ task_info, const uint32_t* __restrict__ group_task_base,
                                                              int NG, uint32_t* __restrict__ len_hist, uint32_t S, int low) {
    msm_set_wave_prio(low);
    __shared__ uint32_t sh[MSM_S + 1];
    if (threadIdx.x <= MSM_S) sh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ntasks = group_task_base[NG];
    for (uint32_t base = blockIdx.x * 8192u; base < ntasks; base += gridDim.x * 8192u) {
        for (uint32_t k = 0; k < 8; k++) {
            uint32_t t = base + k * 1024u + threadIdx.x;
            uint32_t len = 0;
            if (t < ntasks) {
                uint2 ti = task_info[t];
                len = ti.y - ti.x;
            }
            unsigned long long full = __ballot(len == S);
            if (full && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)full) - 1)) atomicAdd(&sh[S], (uint32_t)__popcll(full));
            if (len != 0 && len != S) atomicAdd(&sh[len], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x <= MSM_S && sh[threadIdx.x]) atomicAdd(&len_hist[threadIdx.x], sh[threadIdx.x]);
}
// (the start offset of each length class, longest first, is scanned from len_hist by every workgroup of the scatter below: the class
// cursors count from zero - they are cleared with the other small counters by the sort's single-block launch)
__global__ __launch_bounds__(1024) void msm_len_scatter_kernel(const uint2* __restrict__ task_info,
                                                                 const uint32_t* __restrict__ group_task_base, int NG,
                                                                 const uint32_t* __restrict__ len_hist, uint32_t* __restrict__ len_cursor,
                                                                 uint32_t* __restrict__ order, uint32_t S, int low) {
    msm_set_wave_prio(low);
    __shared__ uint32_t sh_cnt[MSM_S + 1], sh_base[MSM_S + 1], sh_class[MSM_S + 1];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int l = (int)S; l >= 0; l--) {
            sh_class[l] = run;
            run += len_hist[l];
        }
    }
    __syncthreads();
    const uint32_t ntasks = group_task_base[NG];
    for (uint32_t base = blockIdx.x * 1024u; base < ntasks; base += gridDim.x * 1024u) {
        if (threadIdx.x <= MSM_S) sh_cnt[threadIdx.x] = 0;
        __syncthreads();
        uint32_t t = base + threadIdx.x;
        uint32_t len = 0, rank = 0;
        if (t < ntasks) {
            uint2 ti = task_info[t];
            len = ti.y - ti.x;
        }
        unsigned long long full = __ballot(len == S);
        if (len == S) {
            int lane = threadIdx.x & 63, leader = __ffsll((long long)full) - 1;
            uint32_t wbase = 0;
            if (lane == leader) wbase = atomicAdd(&sh_cnt[S], (uint32_t)__popcll(full));
            wbase = __shfl(wbase, leader);
            rank = wbase + (uint32_t)__popcll(full & ((1ull << lane) - 1ull));
        } else if (len != 0) {
            rank = atomicAdd(&sh_cnt[len], 1u);
        }
        __syncthreads();
        if (threadIdx.x <= MSM_S && sh_cnt[threadIdx.x]) sh_base[threadIdx.x] = sh_class[threadIdx.x] + atomicAdd(&len_cursor[threadIdx.x], sh_cnt[threadIdx.x]);
        __syncthreads();
        if (len != 0) order[sh_base[len] + rank] = t;
        __syncthreads();
    }
}

// ---- 4. accumulate -------------------------------------------------------------------------
// The kernel lives in msm_acc.hip: that translation unit is compiled with the multiplier inlined
// (no argument marshalling around the ten products of a mixed addition); everything else in this
// file calls the multiplier as a function to keep the latency-bound tail kernels small.
template <class P>
void msm_launch_accumulate(const uint32_t* sorted, const Affine<P>* table, const uint2* task_info, const uint32_t* order,
                           const uint32_t* group_task_base, int NG, Xyzz<P>* partials, size_t nt, hipStream_t s);
template <class P>
void msm_launch_accumulate_persistent(const uint32_t* sorted, const Affine<P>* table, const uint2* task_info, const uint32_t* order,
                                      const uint32_t* group_task_base, int NG, Xyzz<P>* partials, uint32_t* cursor, hipStream_t s, unsigned wgs_per_cu = 0);

// ---- 4b. the small-commitment path (msm_small.hip): a resident key of <= 2^16 points keeps every multiple of every window base ----
constexpr size_t MSM_SMALL_MAX_POINTS = (size_t)1 << 16;
int msm_small_window_bits(size_t n);
size_t msm_small_table_entries(size_t n, int c);
unsigned msm_small_groups(size_t n, int c);
unsigned msm_small_out_points(unsigned groups);
template <class P>
void msm_small_build_table(const Affine<P>* wbases, size_t n, int c, Affine<P>* table, hipStream_t s);
template <class P, class SF>
void msm_small_launch(const void* d_scalars, size_t n, int is_mont, const Affine<P>* table, int c, void* group_pts, uint32_t* counter, Xyzz<P>* out,
                      hipStream_t s);
size_t msm_small_group_bytes();
size_t msm_small_scratch_bytes();

// ---- 6'. the bucket reduction (msm_reduce.hip): one launch per level of the bit-plane merge tree, radix-2^29 points ----
size_t msm_reduce_plane_bytes(size_t nb);
template <class P>
void msm_launch_reduce(const Xyzz<P>* buckets, void* planes_a, void* planes_b, int c, int G, uint32_t B, Xyzz<P>* out_host, hipStream_t s);

// Switches of the commitments-in-flight path (read once per process; the defaults are the measured best).
// Everything else that round 2 kept for A/B runs (stream / wave priorities off, more waves per SIMD, a 128-VGPR build, hipGraph
// replay, background-behind-sort off) lost its measurement and is gone: the winning setting is now the only code path.
struct MsmTuning {
    int persistent, max_acc, placement_log, bucket_direct;
    size_t persistent_min;
    MsmTuning() {
        auto geti = [](const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; };
        persistent = geti("LURK_MSM_ACC_PERSISTENT", 1);  // DEFAULT-class commitments in flight: 0 = plain launch, 1 = persistent from persistent_min entries, 2 = always
        // W n entries (in 2^20) from which a commitment in flight takes the persistent form.  A one-wave-per-SIMD accumulation runs at
        // ~55 % of the plain launch's rate; two of them resident pay that back only when the accumulation is long against the sort and
        // tail around it: 2^21 scalars x 13 windows (27 M entries) break even, 2^20 (13.6 M) is 4 % faster with the plain launch
        // (853-858 against 821-825 Mscalar-mul/s, two in flight; profiles/r05_persistent_threshold.txt).
        persistent_min = (size_t)geti("LURK_MSM_PERSISTENT_MIN_MENTRIES", 24) << 20;
        max_acc = geti("LURK_MSM_MAX_ACC", 2);            // persistent accumulations resident at once (0 = no limit)
        bucket_direct = geti("LURK_MSM_BUCKET_DIRECT", 1);  // 0: short commitments keep the planned-task stages (A/B runs, parity test)
        placement_log = geti("LURK_MSM_PLACEMENT_LOG", 0);  // diagnostic: persistent workgroups per CU, on stderr
        if (max_acc > 2) max_acc = 0;
    }
};
static const MsmTuning& msm_tuning() {
    static const MsmTuning t;
    return t;
}

// ---- 3-5 in one launch for commitments with few buckets (msm_bucket_direct.hip) ---------------------------------------------------
constexpr uint32_t MSM_DIRECT_MAX_BUCKETS = 131072;          // two key spaces of 16-bit windows
constexpr size_t MSM_DIRECT_MAX_ENTRIES = (size_t)1 << 21;   // W n: 2^16 points x 16 windows, or a pair over 2^17 composed scalars
template <class P>
void msm_launch_bucket_direct(const uint32_t* sorted, const Affine<P>* table, const uint32_t* bucket_start, const uint32_t* cnt, uint32_t NB, size_t entries,
                              Xyzz<P>* buckets, uint32_t* big_list, uint32_t* big_count, hipStream_t s);

// ---- 5. finalize: buckets of <= MSM_SMALL task partials, one lane each (msm_finalize.hip) ----------------------------------
template <class P>
void msm_launch_finalize(const Xyzz<P>* partials, const uint32_t* cnt, const uint32_t* task_start, const uint32_t* group_task_base, uint32_t NB,
                         Xyzz<P>* buckets, uint32_t* big_list, uint32_t* big_count, uint32_t S, hipStream_t s);

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
                                                               const uint32_t* __restrict__ group_task_base, Xyzz<P>* __restrict__ buckets,
                                                               const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count, uint32_t S) {
    msm_set_wave_prio(0);
    extern __shared__ uint4 lds_raw[];
    Xyzz<P>* sh = reinterpret_cast<Xyzz<P>*>(lds_raw);
    const uint32_t nbig = *big_count;
    for (uint32_t i = blockIdx.x; i < nbig; i += gridDim.x) {
        uint32_t key = big_list[i];
        uint32_t g = key / MSM_GRP, b = key % MSM_GRP;
        uint32_t nt = (cnt[key] + S - 1) / S;
        uint32_t first = group_task_base[g] + task_start[(size_t)g * (MSM_GRP + 1) + b];
        Xyzz<P> acc = xyzz_identity<P>();
        for (uint32_t j = threadIdx.x; j < nt; j += 256) xyzz_add<P>(acc, partials[first + j]);
        block_tree_sum<P, 256>(acc, sh);
        if (threadIdx.x == 0) buckets[key] = acc;
        __syncthreads();
    }
}

// ---- 6. bucket reduction: msm_reduce.hip ------------------------------------------------------------------------------------
// ---- precomputed table: T[w*n + i] = 2^(c w) * P_i: msm_precompute.hip ---------------------------
size_t msm_precompute_scratch_bytes(size_t n, int W);
template <class P>
void msm_launch_precompute(const Affine<P>* bases, size_t n, Affine<P>* table, int c, int W, void* scratch, hipStream_t s);

// ---- context -------------------------------------------------------------------------------
constexpr int MSM_SLOTS = LURK_MSM_SLOTS;  // commitments in flight per context (independent workspaces + streams)

struct MsmCtxBase {
    int curve = 0;
    int device = 0;  // the device the context lives on: every entry point runs under a DeviceGuard for it
    size_t npoints = 0;
    bool precomputed = false;
    bool small = false;  // precomputed in the small-commitment form (msm_small.hip): `c` is its window width, no bucket pipeline
    int c = MSM_C_PLAIN;
    bool keep_buffers = false;  // a context whose points are replaced again and again (the inner-product argument's folded key): the table
                                // buffer and the precomputation's scratch stay allocated between set_bases_device calls
    virtual ~MsmCtxBase() {}
    // synchronous: enqueue on `s` with slot 0's workspace, wait, host tail
    virtual void run(const void* d_scalars, size_t n, int is_mont, hipStream_t s, void* out_jac96_host) = 0;
    // asynchronous: enqueue on the slot's own stream (after `after`, the stream that produced the scalars)
    virtual void submit(int slot, const void* d_scalars, size_t n, int is_mont, hipStream_t after, int mode) = 0;
    virtual void wait(int slot, void* out_jac96_host) = 0;
    // two commitments with disjoint supports in one pass: scalars whose index has bit sel_bit clear -> out_lo, set -> out_hi
    virtual void submit_pair(int slot, const void* d_scalars, size_t n, int is_mont, hipStream_t after, int sel_bit) = 0;
    virtual void wait_pair(int slot, void* out_lo_jac96_host, void* out_hi_jac96_host) = 0;
    virtual void wait_pair_xyzz(int slot, void* out_lo_xyzz128_host, void* out_hi_xyzz128_host) = 0;  // not normalised: no field inversion
    // pasta-msm's calling convention: everything in host memory, nothing resident (buffers and workspaces are kept for the next call)
    virtual void run_oneshot(const void* bases, const void* scalars, size_t n, int is_mont, void* out_jac96_host) = 0;
    virtual void rebind(const void* d_bases, size_t n) = 0;  // plain key over other (borrowed) device bases, workspaces kept
    virtual void reserve(size_t n, int slots) = 0;  // allocate the workspaces of slots 0..slots-1 for n scalars now
    virtual const void* device_table() const = 0;  // npoints (x windows when precomputed) 64-byte records
    // adopt a table that is already complete in device memory (loaded from a key file)
    virtual void adopt_table(DevBuf&& buf, size_t n, bool precomputed_, int c_) = 0;
};

// lurk_hip_msm_oneshot_key_cache(1) turns it on (default off); LURK_MSM_ONESHOT_KEY_CACHE=1 in the environment does the same at load
// time, for a pasta-msm drop-in that links the one-shot symbols unchanged and has no way to call that function
static int oneshot_key_cache_env() {
    const char* v = getenv("LURK_MSM_ONESHOT_KEY_CACHE");
    return v && atoi(v) != 0;
}
static std::atomic<int> g_oneshot_key_cache{oneshot_key_cache_env()};
static bool oneshot_key_cache_enabled() { return g_oneshot_key_cache.load() != 0; }

template <class P, class SF>
struct MsmCtx : MsmCtxBase {
    DevBuf own_bases;                // bases (or the whole table when precomputed)
    DevBuf pre_scratch;              // keep_buffers: the table precomputation's scratch
    const Affine<P>* table = nullptr;
    DevBuf small_table;              // small form: n x W x 2^(c-1) multiples (own_bases then holds the plain bases)

    struct Work {
        std::mutex mu;
        DevBuf inter, sorted, block_hist, part_cnt, part_start, cnt, bucket_start, task_start, group_tasks, group_task_base, task_info,
            task_order, len_hist, partials, buckets, big_list, big_count, planes_a, planes_b, canon, small_counter;
        Xyzz<P>* host_pts = nullptr;  // pinned: window sums or bit planes for the host tail
        size_t ws_n = 0, ws_entries = 0, ws_nt = 0;  // what the workspaces hold room for: scalars (0 = nothing yet), sorted entries, tasks,
        uint32_t ws_NB = 0;                          // keys
        hipStream_t stream = nullptr;      // slot stream: sort, plan, finalize, reduce (high priority)
        hipStream_t acc_stream = nullptr;  // the accumulate kernel alone (low priority)
        hipEvent_t ready = nullptr, planned = nullptr, accumulated = nullptr;
        hipEvent_t acc_gate = nullptr;     // LURK_MSM_SUBMIT_FOLLOW: the accumulation waits for this event (the end of the followed commitment's accumulation)
        int follow_wgs = 0;                // LURK_MSM_SUBMIT_FOLLOW: persistent accumulation with this many waves per SIMD (0: the plain launch)
        bool follow_low = false;           // LURK_MSM_SUBMIT_FOLLOW: the commitment's short kernels at the lowest wave priority
        DevBuf cursor;                     // task cursor of the persistent accumulate kernel (+ its per-CU placement counters)
        bool placement_valid = false;
        int sel = -1;                      // the pending commitment is a PAIR split by this bit of the scalar index (submit_pair)
        bool small_ready = false;          // the small path's buffers exist and its arrival counter is zero
        bool force_persistent = false;     // LURK_MSM_SUBMIT_BACKGROUND
        bool foreground = false;           // LURK_MSM_SUBMIT_FOREGROUND
        hipStream_t pending_stream = nullptr;  // the stream the pending commitment ends on
        bool pending = false;
        size_t pending_n = 0;
        ~Work() {
            if (host_pts) (void)hipHostFree(host_pts);
            if (stream) (void)hipStreamDestroy(stream);
            if (acc_stream) (void)hipStreamDestroy(acc_stream);
            if (ready) (void)hipEventDestroy(ready);
            if (planned) (void)hipEventDestroy(planned);
            if (accumulated) (void)hipEventDestroy(accumulated);
        }
    };
    Work work[MSM_SLOTS];
    std::mutex fg_mu;
    Work* last_fg = nullptr;  // the slot of the latest LURK_MSM_SUBMIT_FOREGROUND commitment
    std::mutex acc_ring_mu;  // the last MSM_SLOTS persistent accumulations, in submission order
    Work* acc_ring[MSM_SLOTS] = {};
    int acc_ring_pos = 0, acc_ring_count = 0;

    MsmShape shape(size_t n, int sel = -1) const { return msm_make_shape(c, precomputed, npoints, n, sel); }

    // small_pref: 0 = the small form when it fits a quarter of the free memory (the default), 1 = the small form or an error, -1 = never
    void set_bases_device(const void* d_bases, size_t n, bool copy, bool precompute, int c_override, hipStream_t s, int small_pref = 0) {
        npoints = n;
        precomputed = precompute;
        small = false;
        small_table.release();
        bool small_form = precompute && !c_override && n > 0 && n <= MSM_SMALL_MAX_POINTS && small_pref >= 0;
        LURK_REQUIRE(small_pref <= 0 || small_form, "LURK_MSM_FLAG_SMALL_FORM: the small form needs the precompute flag, no window override and 1 .. 2^16 points");
        if (small_form && small_pref > 0) {
            // the form was asked for by name (LURK_MSM_FLAG_SMALL_FORM): refuse BEFORE allocating gigabytes when the device cannot hold it
            // (table + build scratch), with the out-of-memory code the caller's fallback tests for.  LURK_MSM_SMALL_FORM_MAX_MB caps what a
            // small-form key may take (a deployment knob, and the way the test of the refusal path forces it).
            const int cs = msm_small_window_bits(n);
            const size_t need = msm_small_table_entries(n, cs) * sizeof(Affine<P>) + msm_small_scratch_bytes() + (size_t)msm_num_windows(cs) * n * 160;
            size_t free_b = 0, total_b = 0;
            LURK_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            const char* cap_env = getenv("LURK_MSM_SMALL_FORM_MAX_MB");  // read per call: a deployment may set it between keys
            const size_t cap_mb = cap_env ? (size_t)atoll(cap_env) : (size_t)0;
            if (need > free_b || (cap_mb && need > (cap_mb << 20)))
                throw HipFailure{LURK_HIP_ERR_OOM, "LURK_MSM_FLAG_SMALL_FORM: the small-commitment form of this key needs " + std::to_string(need >> 20) +
                                                       " MiB (free: " + std::to_string(free_b >> 20) + " MiB" + (cap_mb ? ", LURK_MSM_SMALL_FORM_MAX_MB=" + std::to_string(cap_mb) : std::string()) + ")"};
        }
        if (small_form && small_pref == 0) {
            // the small form is a memory-for-latency trade sized for 288 GB: 256 KiB per point resident (4.3 GB at 2^14 points, 5.6 GB
            // at 2^16) + <= 1 GiB of build scratch.  It is taken only when that is at most a quarter of what the device has free
            // right now (several keys per process, slices of a multi-device key and smaller devices then get the window table)
            const int cs = msm_small_window_bits(n);
            const size_t need = msm_small_table_entries(n, cs) * sizeof(Affine<P>) + msm_small_scratch_bytes() + (size_t)msm_num_windows(cs) * n * 160;
            size_t free_b = 0, total_b = 0;
            LURK_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            if (need > free_b / 4) small_form = false;
        }
        if (small_form) {
            // small resident key: all multiples of all window bases (msm_small.hip); the plain bases stay too (key files, rebinding)
            c = msm_small_window_bits(n);
            const int Ws = msm_num_windows(c);
            own_bases.alloc(n * sizeof(Affine<P>));
            LURK_HIP_CHECK(hipMemcpyAsync(own_bases.p, d_bases, n * sizeof(Affine<P>), hipMemcpyDeviceToDevice, s));
            table = own_bases.as<Affine<P>>();
            {
                DevBuf wbases((size_t)Ws * n * sizeof(Affine<P>)), scratch(msm_precompute_scratch_bytes(n, Ws));
                {
                    ProfScope ps("msm_precompute", s);
                    msm_launch_precompute<P>((const Affine<P>*)d_bases, n, wbases.as<Affine<P>>(), c, Ws, scratch.p, s);
                }
                small_table.alloc(msm_small_table_entries(n, c) * sizeof(Affine<P>));
                msm_small_build_table<P>(wbases.as<Affine<P>>(), n, c, small_table.as<Affine<P>>(), s);  // synchronises s
            }
            small = true;
            return;
        }
        // plain: 16-bit windows (W = 16 key spaces of 2^15 buckets).  With the table every window shares
        // one key space, so the window grows to 20 bits (13 windows whose top one still holds 15 bits).
        // (small keys keep 16-bit windows with the table as well: the 2^19 buckets of a 20-bit window cost a fixed 0.4 ms of
        // bucket reduction - 2^13 points: 0.68 -> 0.45 ms, 2^18: 0.93 -> 0.83 ms; from 2^19 points on the 20-bit window wins)
        c = precompute ? (n <= ((size_t)1 << 18) ? 16 : 20) : MSM_C_PLAIN;
        if (c_override) c = c_override;
        LURK_REQUIRE(c >= 16 && c <= 20, "window bits must be in 16..20");
        LURK_REQUIRE(precompute || c == MSM_C_PLAIN, "the plain mode uses 16-bit windows");
        const int W = msm_num_windows(c);
        // sorted entries carry a 31-bit table index (+ sign bit); the sort's offsets are 32-bit over the W n entries
        LURK_REQUIRE((size_t)W * n < ((size_t)1 << 31), "too many points: windows x points must stay below 2^31");
        if (precompute) {
            if (keep_buffers) own_bases.ensure((size_t)W * n * sizeof(Affine<P>));
            else own_bases.alloc((size_t)W * n * sizeof(Affine<P>));
            if (n) {
                DevBuf once;
                DevBuf& scratch = keep_buffers ? pre_scratch : once;
                scratch.ensure(msm_precompute_scratch_bytes(n, W));
                {
                    ProfScope ps("msm_precompute", s);
                    msm_launch_precompute<P>((const Affine<P>*)d_bases, n, own_bases.as<Affine<P>>(), c, W, scratch.p, s);
                }
                LURK_HIP_CHECK(hipStreamSynchronize(s));  // the scratch buffer is released here
            }
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            table = own_bases.as<Affine<P>>();
        } else if (copy) {
            own_bases.alloc(n * sizeof(Affine<P>));
            LURK_HIP_CHECK(hipMemcpyAsync(own_bases.p, d_bases, n * sizeof(Affine<P>), hipMemcpyDeviceToDevice, s));
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            table = own_bases.as<Affine<P>>();
        } else {
            table = (const Affine<P>*)d_bases;  // borrowed
        }
    }

    DevBuf oneshot_scalars;
    // Opt-in key cache of the one-shot entry points (lurk_hip_msm_oneshot_key_cache): the bases of the previous call stay in HBM
    // with ONESHOT_SAMPLES of them kept on the host; a call whose `bases` pointer and (prefix) length match and whose points at the
    // sampled positions are bit-identical skips the 64 B/point upload.  A commitment key is immutable and lives at one address
    // for a whole proof, which is the only caller this is for; a buffer rewritten at unsampled positions only would be missed.
    static constexpr size_t ONESHOT_SAMPLES = 4096;
    const void* oneshot_host = nullptr;
    size_t oneshot_n = 0;
    std::vector<char> oneshot_samples;
    static size_t oneshot_sample_pos(size_t k, size_t n) { return (size_t)(((unsigned __int128)k * n) / ONESHOT_SAMPLES); }
    bool oneshot_key_matches(const void* bases, size_t n) const {
        if (!oneshot_host || bases != oneshot_host || n > oneshot_n || oneshot_samples.empty()) return false;
        const size_t cnt = oneshot_n < ONESHOT_SAMPLES ? oneshot_n : ONESHOT_SAMPLES;
        for (size_t k = 0; k < cnt; k++) {
            const size_t i = oneshot_n < ONESHOT_SAMPLES ? k : oneshot_sample_pos(k, oneshot_n);
            if (i >= n) break;
            if (memcmp((const char*)bases + i * 64, oneshot_samples.data() + k * 64, 64) != 0) return false;
        }
        return true;
    }
    void oneshot_key_remember(const void* bases, size_t n) {
        const size_t cnt = n < ONESHOT_SAMPLES ? n : ONESHOT_SAMPLES;
        oneshot_samples.resize(cnt * 64);
        for (size_t k = 0; k < cnt; k++) memcpy(oneshot_samples.data() + k * 64, (const char*)bases + (n < ONESHOT_SAMPLES ? k : oneshot_sample_pos(k, n)) * 64, 64);
        oneshot_host = bases;
        oneshot_n = n;
    }
    void run_oneshot(const void* bases, const void* scalars, size_t n, int is_mont, void* out) override {
        if (n == 0) {
            put_identity(out);
            return;
        }
        LURK_REQUIRE((size_t)msm_num_windows(MSM_C_PLAIN) * n < ((size_t)1 << 31), "too many points");
        Work& wk = work[0];
        std::lock_guard<std::mutex> lk(wk.mu);
        LURK_REQUIRE(!wk.pending, "slot 0 has a submitted commitment that was not waited for");
        const bool cached = oneshot_key_cache_enabled() && oneshot_key_matches(bases, n);
        if (!cached) {
            oneshot_host = nullptr;  // the device copy is about to be overwritten
            own_bases.ensure(n * sizeof(Affine<P>));
        }
        oneshot_scalars.ensure(n * 32);
        table = own_bases.as<Affine<P>>();
        npoints = n;
        precomputed = false;
        small = false;
        small_table.release();
        c = MSM_C_PLAIN;
        ensure_streams(wk);
        hipStream_t s = wk.stream;
        // scalars first (the sort needs only them); the 64 B/point of bases follow behind the sort, right before the accumulation
        LURK_HIP_CHECK(hipMemcpyAsync(oneshot_scalars.p, scalars, n * 32, hipMemcpyHostToDevice, s));
        const std::function<void()> upload_bases = [&] {
            if (!cached) LURK_HIP_CHECK(hipMemcpyAsync(own_bases.p, bases, n * sizeof(Affine<P>), hipMemcpyHostToDevice, s));
        };
        enqueue(wk, oneshot_scalars.p, n, is_mont, s, nullptr, &upload_bases);
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        if (!cached && oneshot_key_cache_enabled()) oneshot_key_remember(bases, n);
        host_tail(wk, n, out);
    }
    void rebind(const void* d_bases, size_t n) override {
        LURK_REQUIRE((size_t)msm_num_windows(MSM_C_PLAIN) * n < ((size_t)1 << 31), "too many points");
        for (auto& wk : work) {
            std::lock_guard<std::mutex> lk(wk.mu);
            LURK_REQUIRE(!wk.pending, "a slot has a commitment in flight");
        }
        own_bases.release();
        small_table.release();
        small = false;
        table = (const Affine<P>*)d_bases;
        npoints = n;
        precomputed = false;
        c = MSM_C_PLAIN;
    }
    void reserve(size_t n, int slots) override {
        LURK_REQUIRE(n <= npoints, "more scalars than bases in the context");
        LURK_REQUIRE(slots >= 1 && slots <= MSM_SLOTS, "slot count out of range");
        if (n == 0) return;
        if (small) {
            DevBuf zeros(32);
            // (hipMemset of device memory is a NULL-stream operation that may return before it has run, and the NULL stream orders nothing
            // against this library's non-blocking streams: wait for it before a kernel on another stream reads the buffer)
            LURK_HIP_CHECK(hipMemset(zeros.p, 0, 32));
            LURK_HIP_CHECK(hipStreamSynchronize(nullptr));
            for (int k = 0; k < slots; k++) {
                Work& wk = work[k];
                std::lock_guard<std::mutex> lk(wk.mu);
                LURK_REQUIRE(!wk.pending, "slot is busy");
                ensure_streams(wk);
                enqueue(wk, zeros.p, 1, 0, wk.stream, nullptr);  // one launch through the slot's queue: pays its setup now
                LURK_HIP_CHECK(hipStreamSynchronize(wk.stream));
            }
            return;
        }
        const MsmShape sh = shape(n);
        // A slot's first commitment otherwise pays for its two hardware queues and their scratch rings (tens of ms): run one
        // empty commitment (all-zero scalars: no entries, every kernel launched) through each slot's streams now.
        const size_t nz = n < 256 ? n : 256;
        DevBuf zeros(nz * 32);
        LURK_HIP_CHECK(hipMemset(zeros.p, 0, nz * 32));
        LURK_HIP_CHECK(hipStreamSynchronize(nullptr));  // (see above)
        for (int k = 0; k < slots; k++) {
            Work& wk = work[k];
            std::lock_guard<std::mutex> lk(wk.mu);
            LURK_REQUIRE(!wk.pending, "slot is busy");
            ensure_workspace(wk, sh);
            ensure_streams(wk);
            enqueue(wk, zeros.p, nz, 0, wk.stream, nullptr);  // (the plain launch: a slot's low-priority accumulate stream is made when a commitment first needs it)
            LURK_HIP_CHECK(hipStreamSynchronize(wk.stream));
        }
    }
    const void* device_table() const override { return table; }
    void adopt_table(DevBuf&& buf, size_t n, bool precomputed_, int c_) override {
        own_bases = std::move(buf);
        table = own_bases.as<Affine<P>>();
        small_table.release();
        small = false;
        npoints = n;
        precomputed = precomputed_;
        c = c_;
    }

    void ensure_workspace_small(Work& wk, hipStream_t s) {
        if (wk.small_ready) return;
        wk.partials.ensure(msm_small_group_bytes());
        wk.small_counter.ensure(16);
        // the small kernel's arrival counter: zero before its first launch, left zero by every launch.  Zeroed ON THE STREAM of that
        // first launch (round 6): a NULL-stream hipMemset may still be pending when a kernel on a non-blocking stream starts - nothing
        // orders the two - and with a garbage ticket base no workgroup is ever "the last to arrive": the slot's pinned result keeps
        // whatever it held.
        LURK_HIP_CHECK(hipMemsetAsync(wk.small_counter.p, 0, 16, s));
        if (!wk.host_pts) LURK_HIP_CHECK(hipHostMalloc((void**)&wk.host_pts, (size_t)MSM_MAX_W * 20 * sizeof(Xyzz<P>)));
        wk.small_ready = true;
    }

    size_t ntask_max(const MsmShape& sh) const { return (size_t)sh.NB + (size_t)sh.W * sh.n / sh.S + 1; }

    void ensure_workspace(Work& wk, const MsmShape& sh) {
        // the buffer sizes depend on (W n, NB), not on n alone: a context rebound from a table key (c = 20: 13 n entries, 2^19
        // keys) to a plain one (c = 16: 16 n entries, 16 x 2^15 keys) keeps its workspaces and must grow them
        const size_t entries = (size_t)sh.W * sh.n, nt = ntask_max(sh);
        if (wk.ws_n != 0 && entries <= wk.ws_entries && nt <= wk.ws_nt && sh.NB <= wk.ws_NB) return;
        wk.inter.ensure(msm_inter_bytes(entries));
        wk.sorted.ensure(entries * 4);
        wk.block_hist.ensure((size_t)MSM_NB1 * MSM_P_MAX * 4);
        wk.part_cnt.ensure(MSM_P_MAX * 4);
        wk.part_start.ensure((MSM_P_MAX + 1) * 4);
        wk.cnt.ensure((size_t)sh.NB * 4);
        wk.bucket_start.ensure((size_t)sh.NB * 4);
        wk.task_start.ensure((size_t)sh.NG * (MSM_GRP + 1) * 4);
        wk.group_tasks.ensure(64 * 4);
        wk.group_task_base.ensure(64 * 4);
        wk.task_info.ensure(nt * sizeof(uint2));
        wk.task_order.ensure(nt * 4);
        wk.len_hist.ensure(2 * (MSM_S + 1) * 4);
        wk.partials.ensure(nt * sizeof(Xyzz<P>));
        wk.buckets.ensure((size_t)sh.NB * sizeof(Xyzz<P>));
        wk.big_list.ensure((size_t)sh.NB * 4);
        wk.big_count.ensure(16);
        wk.cursor.ensure(4 * (MSM_PLACEMENT_BASE + 512));
        wk.planes_a.ensure(msm_reduce_plane_bytes(sh.NB));  // level k holds (B >> (k+1)) * (k+2) <= B points per space
        wk.planes_b.ensure(msm_reduce_plane_bytes(sh.NB));
        if (!wk.host_pts) LURK_HIP_CHECK(hipHostMalloc((void**)&wk.host_pts, (size_t)MSM_MAX_W * 20 * sizeof(Xyzz<P>)));
        wk.ws_n = sh.n > wk.ws_n ? sh.n : wk.ws_n;
        wk.ws_entries = entries > wk.ws_entries ? entries : wk.ws_entries;
        wk.ws_nt = nt > wk.ws_nt ? nt : wk.ws_nt;
        wk.ws_NB = sh.NB > wk.ws_NB ? sh.NB : wk.ws_NB;
    }

    // every kernel of one commitment + the D2H of its <= 20 result points, on stream s
    // s_acc: stream of the accumulate kernel (nullptr: same stream, classic launch)
    void enqueue(Work& wk, const void* d_scalars, size_t n, int is_mont, hipStream_t s, hipStream_t s_acc = nullptr,
                 const std::function<void()>* before_accumulate = nullptr) {
        if (small) {  // one launch; <= 16 points land in the slot's pinned buffer
            ensure_workspace_small(wk, s);
            msm_small_launch<P, SF>(d_scalars, n, is_mont, small_table.as<Affine<P>>(), c, wk.partials.p,
                                    wk.small_counter.template as<uint32_t>(), wk.host_pts, s);
            if (wk.planned) LURK_HIP_CHECK(hipEventRecord(wk.planned, s));
            return;
        }
        MsmShape sh = shape(n, wk.sel);
        // LURK_MSM_SUBMIT_FOLLOW: the sort, the plan and the accumulation at the lowest wave priority, the tail (finalize, bucket reduction)
        // at the raised one (see submit_impl)
        const int low = wk.follow_low ? 1 : 0;
        sh.low_prio = low;
        ensure_workspace(wk, sh);
        const size_t nt = ntask_max(sh);
        {
            // canonical scalars (when they arrive in Montgomery form) + the two-pass sort: msm_sort.hip
            if (is_mont) wk.canon.ensure(n * 32);
            ProfScope ps("msm_sort", s);
            MsmSortBufs sb;
            sb.inter = wk.inter.p;
            sb.sorted = wk.sorted.template as<uint32_t>();
            sb.block_hist = wk.block_hist.template as<uint32_t>();
            sb.part_cnt = wk.part_cnt.template as<uint32_t>();
            sb.part_start = wk.part_start.template as<uint32_t>();
            sb.cnt = wk.cnt.template as<uint32_t>();
            sb.bucket_start = wk.bucket_start.template as<uint32_t>();
            sb.canon = wk.canon.p;
            sb.zero[0] = wk.big_count.template as<uint32_t>(); sb.zero_n[0] = 1;
            sb.zero[1] = wk.len_hist.template as<uint32_t>(); sb.zero_n[1] = 2 * (MSM_S + 1);
            sb.zero[2] = wk.cursor.template as<uint32_t>(); sb.zero_n[2] = MSM_PLACEMENT_BASE + 512;
            msm_launch_sort<SF>(sh, d_scalars, is_mont, sb, s);
        }
        const MsmTuning& tn = msm_tuning();
        // few buckets, few entries (a key of <= 2^16 points under 16-bit windows): plan, accumulate and finalize as ONE launch, a few
        // lanes per bucket (msm_bucket_direct.hip)
        // (LURK_MSM_ACC_PERSISTENT=2 - "always the persistent form" - and background submissions keep the planned stages)
        const bool direct = tn.bucket_direct && !wk.force_persistent && !(s_acc && tn.persistent == 2) && sh.NB <= MSM_DIRECT_MAX_BUCKETS &&
                            (size_t)sh.W * sh.n <= MSM_DIRECT_MAX_ENTRIES;
        if (direct) {
            if (before_accumulate) (*before_accumulate)();
            if (wk.planned) LURK_HIP_CHECK(hipEventRecord(wk.planned, s));
            {
                ProfScope ps("msm_accumulate_direct", s);
                msm_launch_bucket_direct<P>(wk.sorted.template as<uint32_t>(), table, wk.bucket_start.template as<uint32_t>(), wk.cnt.template as<uint32_t>(),
                                            sh.NB, (size_t)sh.W * sh.n, wk.buckets.template as<Xyzz<P>>(), wk.big_list.template as<uint32_t>(),
                                            wk.big_count.template as<uint32_t>(), s);
            }
            {
                ProfScope ps("msm_reduce", s);
                msm_launch_reduce<P>(wk.buckets.template as<Xyzz<P>>(), wk.planes_a.p, wk.planes_b.p, sh.c, sh.G, sh.B, wk.host_pts, s);
            }
            LURK_HIP_CHECK(hipGetLastError());
            return;
        }
        {
            ProfScope ps("msm_tasks", s);
            uint32_t* lh = wk.len_hist.template as<uint32_t>();
            hipLaunchKernelGGL(msm_taskscan_kernel, dim3(sh.NG), dim3(1024), 0, s, wk.cnt.template as<uint32_t>(),
                               wk.task_start.template as<uint32_t>(), wk.group_tasks.template as<uint32_t>(), (uint32_t)sh.S, low);
            hipLaunchKernelGGL(msm_tasks_kernel, dim3(div_up(nt, 256)), dim3(256), 0, s, wk.cnt.template as<uint32_t>(),
                               wk.bucket_start.template as<uint32_t>(), wk.task_start.template as<uint32_t>(), wk.group_tasks.template as<uint32_t>(),
                               wk.group_task_base.template as<uint32_t>(), sh.NG, wk.task_info.template as<uint2>(), (uint32_t)sh.S, low);
            hipLaunchKernelGGL(msm_len_hist_kernel, dim3(256), dim3(1024), 0, s, wk.task_info.template as<uint2>(),
                               wk.group_task_base.template as<uint32_t>(), sh.NG, lh, (uint32_t)sh.S, low);
            hipLaunchKernelGGL(msm_len_scatter_kernel, dim3(512), dim3(1024), 0, s, wk.task_info.template as<uint2>(),
                               wk.group_task_base.template as<uint32_t>(), sh.NG, lh, lh + MSM_S + 1, wk.task_order.template as<uint32_t>(), (uint32_t)sh.S, low);
        }
        if (before_accumulate) (*before_accumulate)();  // the one-shot entry point uploads the bases here, behind the sort
        // large commitments in flight take the persistent form on the slot's low-priority accumulate stream (below persistent_min
        // entries the plain launch: see msm_tuning); synchronous calls keep the plain launch
        const bool follow_persistent = wk.follow_wgs > 0;  // LURK_MSM_SUBMIT_FOLLOW: a persistent accumulation of follow_wgs waves per SIMD on the slot stream
        const bool persistent = follow_persistent || (s_acc && (wk.force_persistent || (tn.persistent == 1 ? (size_t)sh.W * sh.n >= tn.persistent_min : tn.persistent != 0)));
        if (wk.planned) LURK_HIP_CHECK(hipEventRecord(wk.planned, s));  // sort and plan are enqueued: a background commitment may start behind this point
        if (wk.acc_gate) LURK_HIP_CHECK(hipStreamWaitEvent(s, wk.acc_gate, 0));
        if (follow_persistent) {
            // the registers two waves per SIMD leave (512 - 2 x 176) hold the waves of what the open step's serial chain launches
            // meanwhile - the reduction levels of commit(T), the folds, the next cross term - where a plain launch (three waves, 504
            // registers) made every one of them queue for an accumulate wave to retire
            wk.placement_valid = true;
            {
                ProfScope ps("msm_accumulate_persistent", s);
                msm_launch_accumulate_persistent<P>(wk.sorted.template as<uint32_t>(), table, wk.task_info.template as<uint2>(),
                                                    wk.task_order.template as<uint32_t>(), wk.group_task_base.template as<uint32_t>(), sh.NG,
                                                    wk.partials.template as<Xyzz<P>>(), wk.cursor.template as<uint32_t>(), s, (unsigned)wk.follow_wgs);
            }
            LURK_HIP_CHECK(hipEventRecord(wk.accumulated, s));
        } else if (persistent) {
            wk.placement_valid = true;  // (the cursor block was zeroed by msm_part_start_kernel)
            LURK_HIP_CHECK(hipEventRecord(wk.planned, s));
            LURK_HIP_CHECK(hipStreamWaitEvent(s_acc, wk.planned, 0));
            if (tn.max_acc >= 1) {
                // at most max_acc accumulations resident at once: this one also waits for the one submitted max_acc submissions ago.
                // Two one-wave-per-SIMD accumulations saturate the VALU and leave every SIMD the registers a 1024-thread sort
                // workgroup of the NEXT commitment needs; a third would take them (measured: the sort then waits for an
                // accumulation to end and the remaining one runs alone at half speed).
                std::lock_guard<std::mutex> lk(acc_ring_mu);
                if (acc_ring_count >= tn.max_acc) {
                    Work* gate = acc_ring[(acc_ring_pos + MSM_SLOTS - tn.max_acc) % MSM_SLOTS];
                    if (gate && gate != &wk && gate->accumulated) LURK_HIP_CHECK(hipStreamWaitEvent(s_acc, gate->accumulated, 0));
                }
                acc_ring[acc_ring_pos % MSM_SLOTS] = &wk;
                acc_ring_pos++;
                if (acc_ring_count < MSM_SLOTS) acc_ring_count++;
            }
            {
                ProfScope ps("msm_accumulate_persistent", s_acc);  // (prefix "msm_accumulate" still matches both forms)
                msm_launch_accumulate_persistent<P>(wk.sorted.template as<uint32_t>(), table, wk.task_info.template as<uint2>(),
                                                    wk.task_order.template as<uint32_t>(), wk.group_task_base.template as<uint32_t>(), sh.NG,
                                                    wk.partials.template as<Xyzz<P>>(), wk.cursor.template as<uint32_t>(), s_acc);
            }
            LURK_HIP_CHECK(hipEventRecord(wk.accumulated, s_acc));
            LURK_HIP_CHECK(hipStreamWaitEvent(s, wk.accumulated, 0));
        } else {
            {
                ProfScope ps("msm_accumulate", s);
                msm_launch_accumulate<P>(wk.sorted.template as<uint32_t>(), table, wk.task_info.template as<uint2>(),
                                         wk.task_order.template as<uint32_t>(), wk.group_task_base.template as<uint32_t>(), sh.NG,
                                         wk.partials.template as<Xyzz<P>>(), nt, s);
            }
            if (wk.accumulated) LURK_HIP_CHECK(hipEventRecord(wk.accumulated, s));  // what a LURK_MSM_SUBMIT_FOLLOW commitment starts behind
        }
        {
            ProfScope ps("msm_finalize", s);
            msm_launch_finalize<P>(wk.partials.template as<Xyzz<P>>(), wk.cnt.template as<uint32_t>(), wk.task_start.template as<uint32_t>(),
                                   wk.group_task_base.template as<uint32_t>(), sh.NB, wk.buckets.template as<Xyzz<P>>(),
                                   wk.big_list.template as<uint32_t>(), wk.big_count.template as<uint32_t>(), (uint32_t)sh.S, s);
            hipLaunchKernelGGL((msm_big_bucket_kernel<P>), dim3(128), dim3(256), 256 * sizeof(Xyzz<P>), s, wk.partials.template as<Xyzz<P>>(),
                               wk.cnt.template as<uint32_t>(), wk.task_start.template as<uint32_t>(),
                               wk.group_task_base.template as<uint32_t>(), wk.buckets.template as<Xyzz<P>>(),
                               wk.big_list.template as<uint32_t>(), wk.big_count.template as<uint32_t>(), (uint32_t)sh.S);
        }
        {
            // the c - 1 levels of the bit-plane merge tree; the last one stores the G x c plane sums into the slot's pinned buffer: the
            // host's Horner over them (G c doublings and as many additions) is cheaper than a 20-deep dependent chain on one lane
            ProfScope ps("msm_reduce", s);
            msm_launch_reduce<P>(wk.buckets.template as<Xyzz<P>>(), wk.planes_a.p, wk.planes_b.p, sh.c, sh.G, sh.B, wk.host_pts, s);
        }
        LURK_HIP_CHECK(hipGetLastError());
    }

    // host tail over the <= 20 points the device left in pinned memory (stream already synchronised)
    // results go to caller memory without an alignment promise: memcpy
    static void put_point(void* out, const Jacobian<P>& j) { memcpy(out, &j, sizeof(j)); }
    static void put_identity(void* out) { put_point(out, jacobian_from_affine<P>(Affine<P>{fe_zero<P>(), fe_zero<P>()})); }
    void host_tail(Work& wk, size_t n, void* out) {
        if (small) {
            const unsigned K = msm_small_out_points(msm_small_groups(n, c));
            Xyzz<P> total = xyzz_identity<P>();
            for (unsigned k = 0; k < K; k++) xyzz_add<P>(total, wk.host_pts[k]);
            put_point(out, jacobian_from_affine<P>(xyzz_to_affine<P>(total)));
            return;
        }
        const MsmShape sh = shape(n);
        const Xyzz<P> total = msm_planes_horner_windows<P>(wk.host_pts, sh.G, sh.c);  // sum_g 2^(c g) (S_g + sum_k 2^k P_gk)
        put_point(out, jacobian_from_affine<P>(xyzz_to_affine<P>(total)));
    }
    // a pair: key space g of the two holds the commitment of the scalars whose index has the selector bit = g
    void host_tail_pair(Work& wk, void* out_lo, void* out_hi) {
        void* outs[2] = {out_lo, out_hi};
        for (int g = 0; g < 2; g++) {
            const Xyzz<P> t = msm_planes_horner_windows<P>(wk.host_pts + (size_t)g * c, 1, c);
            put_point(outs[g], jacobian_from_affine<P>(xyzz_to_affine<P>(t)));
        }
    }

    void run(const void* d_scalars, size_t n, int is_mont, hipStream_t s, void* out_jac96_host) override {
        LURK_REQUIRE(n <= npoints, "more scalars than bases in the context");
        void* out = out_jac96_host;
        if (n == 0) {
            put_identity(out);
            return;
        }
        Work& wk = work[0];
        std::lock_guard<std::mutex> lk(wk.mu);
        LURK_REQUIRE(!wk.pending, "slot 0 has a submitted commitment that was not waited for");
        wk.sel = -1;
        enqueue(wk, d_scalars, n, is_mont, s);
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        host_tail(wk, n, out);
    }

    void submit_pair(int slot, const void* d_scalars, size_t n, int is_mont, hipStream_t after, int sel_bit) override {
        LURK_REQUIRE(precomputed && !small, "a pair of commitments in one pass needs the window-table form of a key (precompute flag, more than 2^16 points "
                                            "or a window-bit override)");
        LURK_REQUIRE(sel_bit >= 0 && sel_bit < 31, "selector bit out of range");
        LURK_REQUIRE(n > 0, "empty pair");
        submit_impl(slot, d_scalars, n, is_mont, after, LURK_MSM_SUBMIT_FOREGROUND, sel_bit);
    }
    void wait_pair(int slot, void* out_lo, void* out_hi) override {
        LURK_REQUIRE(slot >= 0 && slot < MSM_SLOTS, "slot out of range");
        Work& wk = work[slot];
        std::lock_guard<std::mutex> lk(wk.mu);
        LURK_REQUIRE(wk.pending && wk.sel >= 0, "no pair was submitted on this slot");
        wk.pending = false;
        LURK_HIP_CHECK(hipStreamSynchronize(wk.pending_stream ? wk.pending_stream : wk.stream));
        host_tail_pair(wk, out_lo, out_hi);
        wk.sel = -1;
    }
    void wait_pair_xyzz(int slot, void* out_lo, void* out_hi) override {
        LURK_REQUIRE(slot >= 0 && slot < MSM_SLOTS, "slot out of range");
        Work& wk = work[slot];
        std::lock_guard<std::mutex> lk(wk.mu);
        LURK_REQUIRE(wk.pending && wk.sel >= 0, "no pair was submitted on this slot");
        wk.pending = false;
        LURK_HIP_CHECK(hipStreamSynchronize(wk.pending_stream ? wk.pending_stream : wk.stream));
        void* outs[2] = {out_lo, out_hi};
        for (int g = 0; g < 2; g++) {
            const Xyzz<P> t = msm_planes_horner_windows<P>(wk.host_pts + (size_t)g * c, 1, c);
            memcpy(outs[g], &t, sizeof(t));
        }
        wk.sel = -1;
    }
    void submit(int slot, const void* d_scalars, size_t n, int is_mont, hipStream_t after, int mode) override {
        submit_impl(slot, d_scalars, n, is_mont, after, mode, -1);
    }
    void submit_impl(int slot, const void* d_scalars, size_t n, int is_mont, hipStream_t after, int mode, int sel_bit) {
        LURK_REQUIRE(slot >= 0 && slot < MSM_SLOTS, "slot out of range");
        LURK_REQUIRE(n <= npoints, "more scalars than bases in the context");
        Work& wk = work[slot];
        std::lock_guard<std::mutex> lk(wk.mu);
        LURK_REQUIRE(!wk.pending, "slot is busy: wait for it first");
        wk.sel = sel_bit;
        ensure_streams(wk);
        if (n) {
            // the scalars were produced on the caller's stream: order the slot stream after it
            // foreground: the accumulation too on the slot's high-priority stream, more waves per SIMD, raised wave priority;
            // background: persistent one-wave accumulation whatever the size, started behind the foreground commitment's sort.
            // (Its short kernels stay on the high-priority stream: on the low-priority queue every one of the ~45 dependent
            // launches of a commitment was dispatched 40 us late - 50-60 us per bit-plane level instead of 13 - even on an idle chip.)
            const bool bg = mode == LURK_MSM_SUBMIT_BACKGROUND, follow = mode == LURK_MSM_SUBMIT_FOLLOW;
            wk.foreground = mode == LURK_MSM_SUBMIT_FOREGROUND || follow;
            wk.force_persistent = bg;
            wk.pending_stream = wk.stream;
            LURK_HIP_CHECK(hipEventRecord(wk.ready, after));
            LURK_HIP_CHECK(hipStreamWaitEvent(wk.pending_stream, wk.ready, 0));
            {
                // a background commitment starts behind the sort of the foreground commitment in flight: its own sort (LDS atomics,
                // barriers) then runs under the foreground accumulation (integer VALU) instead of beside the foreground sort
                std::lock_guard<std::mutex> lk2(fg_mu);
                if (bg && last_fg && last_fg != &wk && last_fg->planned)
                    LURK_HIP_CHECK(hipStreamWaitEvent(wk.pending_stream, last_fg->planned, 0));
                // follow (round 6): work staged ahead that must only take what the open step's serial chain leaves.  Its sort and plan
                // run at once, at the LOWEST wave priority (beside the cross term and commit(T)'s own sort: they only have to be through
                // when commit(T)'s accumulation ends); its accumulation waits for the END of the followed commitment's accumulation
                // (the event that commitment's enqueue recorded: the caller's thread has returned from it; an event of an earlier,
                // finished commitment - or one never recorded - orders nothing) and is the persistent form with follow_wgs waves per
                // SIMD, lowest priority: what two waves leave of a SIMD's registers (512 - 2 x 176) holds the waves of commit(T)'s
                // reduction levels, the folds and the next cross term, where a plain launch (three waves, 504 registers) made each of
                // them queue for an accumulate wave to retire (measured: a 60 us fold 320 us, the reduction 0.35 -> 0.70 ms).  Its
                // TAIL keeps the raised priority: the next step's begin waits for this commitment as well as for its own commit(T).
                wk.acc_gate = nullptr;
                wk.follow_wgs = 0;
                wk.follow_low = false;
                if (follow) {
                    static const int wgs = [] { const char* v = getenv("LURK_MSM_FOLLOW_WGS"); const int x = v ? atoi(v) : 2; return x < 0 ? 0 : x > 3 ? 3 : x; }();
                    wk.follow_wgs = wgs;  // 0: the plain launch (A/B runs)
                    wk.follow_low = true;
                    if (last_fg && last_fg != &wk && last_fg->accumulated) wk.acc_gate = last_fg->accumulated;
                }
                if (mode == LURK_MSM_SUBMIT_FOREGROUND) last_fg = &wk;
            }
            // foreground: everything on the (high-priority) slot stream
            hipStream_t acc_s = wk.foreground ? nullptr : ensure_acc_stream(wk);
            enqueue(wk, d_scalars, n, is_mont, wk.pending_stream, acc_s);
            wk.acc_gate = nullptr;
            wk.follow_wgs = 0;
            wk.follow_low = false;
        }
        wk.pending = true;
        wk.pending_n = n;
    }

    // The low-priority stream of a slot's persistent accumulation, made on first need (round 6): a stream of a new priority class costs
    // the process another pool of hardware queues (GPU_MAX_HW_QUEUES per class), and a prover whose commitments are all FOREGROUND /
    // FOLLOW (the folding step) never launches on it - its streams then share the queues of two classes instead of three.
    hipStream_t ensure_acc_stream(Work& wk) {
        if (!wk.acc_stream) {
            int least = 0, greatest = 0;
            LURK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            LURK_HIP_CHECK(hipStreamCreateWithPriority(&wk.acc_stream, hipStreamNonBlocking, least));
        }
        return wk.acc_stream;
    }

    void ensure_streams(Work& wk) {
        if (!wk.stream) {
            // the throughput-bound accumulate kernel runs on its own low-priority stream, everything else of the slot on a
            // high-priority one: the short kernels of the next commitment are dispatched ahead of it as wave slots free up
            int least = 0, greatest = 0;
            LURK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            LURK_HIP_CHECK(hipStreamCreateWithPriority(&wk.stream, hipStreamNonBlocking, greatest));
            LURK_HIP_CHECK(hipEventCreateWithFlags(&wk.ready, hipEventDisableTiming));
            LURK_HIP_CHECK(hipEventCreateWithFlags(&wk.planned, hipEventDisableTiming));
            LURK_HIP_CHECK(hipEventCreateWithFlags(&wk.accumulated, hipEventDisableTiming));
        }
    }

    void wait(int slot, void* out_jac96_host) override {
        LURK_REQUIRE(slot >= 0 && slot < MSM_SLOTS, "slot out of range");
        Work& wk = work[slot];
        std::lock_guard<std::mutex> lk(wk.mu);
        LURK_REQUIRE(wk.pending, "nothing was submitted on this slot");
        LURK_REQUIRE(wk.sel < 0, "a pair is pending on this slot: use lurk_hip_msm_ctx_wait_pair");
        void* out = out_jac96_host;
        wk.pending = false;
        if (wk.pending_n == 0) {
            put_identity(out);
            return;
        }
        LURK_HIP_CHECK(hipStreamSynchronize(wk.pending_stream ? wk.pending_stream : wk.stream));
        if (wk.placement_valid && msm_tuning().placement_log) log_placement(wk);
        host_tail(wk, wk.pending_n, out);
    }
    // LURK_MSM_PLACEMENT_LOG=1: how the dispatcher spread the persistent accumulate's workgroups (one per CU wanted)
    void log_placement(Work& wk) {
        std::vector<uint32_t> h(MSM_PLACEMENT_BASE + 512);
        LURK_HIP_CHECK(hipMemcpy(h.data(), wk.cursor.p, h.size() * 4, hipMemcpyDeviceToHost));
        int hist[8] = {0}, used = 0;
        for (int i = 0; i < 512; i++) {
            uint32_t c = h[MSM_PLACEMENT_BASE + i];
            if (c) used++;
            hist[c > 7 ? 7 : c]++;
        }
        fprintf(stderr, "[lurk_hip] accumulate placement: %d CUs used; CUs with 1/2/3/4+ workgroups: %d/%d/%d/%d\n", used, hist[1], hist[2], hist[3],
                hist[4] + hist[5] + hist[6] + hist[7]);
        wk.placement_valid = false;
    }
};

static MsmCtxBase* new_ctx(int curve) {
    LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
    MsmCtxBase* c = curve == LURK_CURVE_PALLAS ? (MsmCtxBase*)new MsmCtx<PallasFp, PallasFq>() : (MsmCtxBase*)new MsmCtx<PallasFq, PallasFp>();
    c->curve = curve;
    c->device = current_device();
    return c;
}
static void ctx_set_bases(MsmCtxBase* c, const void* d_bases, size_t n, bool copy, int flags, hipStream_t s) {
    bool pre = (flags & LURK_MSM_FLAG_PRECOMPUTE) != 0;
    int c_override = (flags >> 8) & 0xff;
    LURK_REQUIRE(!((flags & LURK_MSM_FLAG_SMALL_FORM) && (flags & LURK_MSM_FLAG_NO_SMALL_FORM)), "LURK_MSM_FLAG_SMALL_FORM and LURK_MSM_FLAG_NO_SMALL_FORM exclude each other");
    const int small_pref = (flags & LURK_MSM_FLAG_SMALL_FORM) ? 1 : (flags & LURK_MSM_FLAG_NO_SMALL_FORM) ? -1 : 0;
    if (c->curve == LURK_CURVE_PALLAS) static_cast<MsmCtx<PallasFp, PallasFq>*>(c)->set_bases_device(d_bases, n, copy, pre, c_override, s, small_pref);
    else static_cast<MsmCtx<PallasFq, PallasFp>*>(c)->set_bases_device(d_bases, n, copy, pre, c_override, s, small_pref);
}

template <class P>
static void point_sum_host(const void* pts, size_t count, void* out) {
    // the caller's buffers carry no alignment promise (Fe<P> is 16-byte aligned): go through memcpy
    Xyzz<P> acc = xyzz_identity<P>();
    for (size_t i = 0; i < count; i++) {
        Jacobian<P> j;
        memcpy(&j, (const char*)pts + i * sizeof(Jacobian<P>), sizeof(j));
        xyzz_add<P>(acc, xyzz_from_jacobian<P>(j));
    }
    const Jacobian<P> r = jacobian_from_affine<P>(xyzz_to_affine<P>(acc));
    memcpy(out, &r, sizeof(r));
}
template <class P>
static void point_affine_canonical_host(const void* pt, void* out) {
    Jacobian<P> j;
    memcpy(&j, pt, sizeof(j));
    // every commitment this library hands out is normalised (Z = the Montgomery one): then (X, Y) ARE the affine coordinates and the field
    // inversion (~17 us on a host core: a quarter of the transcript's time when four commitments are absorbed) is skipped
    const Fe<P> one = fe_one<P>();
    bool z_is_one = true;
    for (int i = 0; i < 8; i++) z_is_one = z_is_one && j.z.l[i] == one.l[i];
    Affine<P> a = z_is_one ? Affine<P>{j.x, j.y} : xyzz_to_affine<P>(xyzz_from_jacobian<P>(j));
    Fe<P> x = fe_from_mont<P>(a.x), y = fe_from_mont<P>(a.y);
    memcpy(out, x.l, 32);
    memcpy((char*)out + 32, y.l, 32);
}

// [k] P on the host (double-and-add over the canonical scalar, top bit first): a handful per folding step
template <class P, class SF>
static void point_mul_host(const void* pt, const void* scalar32, int is_mont, void* out) {
    Fe<SF> k;
    memcpy(k.l, scalar32, 32);
    if (is_mont) k = fe_from_mont<SF>(k);
    Jacobian<P> j;
    memcpy(&j, pt, sizeof(j));
    const Xyzz<P> base = xyzz_from_jacobian<P>(j);
    Xyzz<P> acc = xyzz_identity<P>();
    int top = 255;  // Nova's folding challenges are 128 bits (NUM_CHALLENGE_BITS): start at the highest set bit
    while (top >= 0 && !((k.l[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int i = top; i >= 0; i--) {
        acc = xyzz_dbl<P>(acc);
        if ((k.l[i >> 5] >> (i & 31)) & 1u) xyzz_add<P>(acc, base);
    }
    const Jacobian<P> r = jacobian_from_affine<P>(xyzz_to_affine<P>(acc));
    memcpy(out, &r, sizeof(r));
}

static int msm_oneshot(int curve, void* out, const void* bases, size_t n, const void* scalars, int is_mont) {
    return guarded([&] {
        LURK_REQUIRE(out, "null output");
        LURK_REQUIRE(n == 0 || (bases && scalars), "null buffer");
        // one cached context per (device, curve): its device buffers and workspaces survive between calls, so an unmodified caller
        // of the pasta-msm symbols pays PCIe (96 B per point) but no allocation.  Calls on one (device, curve) serialise.
        static std::mutex mu;
        static std::map<std::pair<int, int>, std::unique_ptr<MsmCtxBase>> cache;
        MsmCtxBase* c;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto key = std::make_pair(current_device(), curve);
            auto it = cache.find(key);
            if (it == cache.end()) it = cache.emplace(key, std::unique_ptr<MsmCtxBase>(new_ctx(curve))).first;
            c = it->second.get();
        }
        c->run_oneshot(bases, scalars, n, is_mont, out);
    });
}

}  // namespace lurk

using namespace lurk;

struct lurk_hip_msm_ctx {
    std::unique_ptr<MsmCtxBase> impl;
};

// A commitment key cut into contiguous slices, one per device of the list, driven from ONE host process:
// each slice has its own context (resident in that device's HBM) and its own host thread bound to the device, so
// the devices sort / accumulate concurrently; the 96-byte partial commitments come back to the host and are
// summed with the host group law.  No bucket array ever crosses a link.
struct lurk_hip_msm_multi {
    struct Shard {
        size_t lo = 0, hi = 0;
        std::unique_ptr<DeviceWorker> worker;
        std::unique_ptr<MsmCtxBase> ctx;  // created, used and destroyed on the worker thread
        DevBuf staged;                    // device copy of this shard's scalars (host-pointer commits)
        Jacobian<PallasFp> partial;       // both curves share the 96-byte layout
    };
    int curve = 0;
    size_t npoints = 0;
    std::vector<std::unique_ptr<Shard>> shards;
    std::mutex mu;  // one commitment at a time per multi-context
    size_t pending_n[MSM_SLOTS] = {};  // asynchronous form: scalars of the commitment in flight on each slot
    bool pending[MSM_SLOTS] = {};

    // f(shard, count) on every shard that owns some of the first n scalars; waits for all of them
    template <class F>
    void for_shards(size_t n, F&& f) {
        std::vector<Shard*> live;
        for (auto& sp : shards) {
            Shard& sh = *sp;
            if (sh.lo >= n || sh.lo == sh.hi) continue;
            size_t cnt = (sh.hi < n ? sh.hi : n) - sh.lo;
            Shard* shp = &sh;
            sh.worker->post([shp, cnt, &f] { f(*shp, cnt); });
            live.push_back(shp);
        }
        std::unique_ptr<HipFailure> first;
        for (Shard* shp : live) {
            try {
                shp->worker->wait();
            } catch (const HipFailure& e) {
                if (!first) first.reset(new HipFailure(e));
            }
        }
        if (first) throw *first;
    }
    void sum(size_t n, void* out) {
        std::vector<Jacobian<PallasFp>> parts;
        for (auto& sp : shards)
            if (sp->lo < n && sp->lo != sp->hi) parts.push_back(sp->partial);
        if (curve == LURK_CURVE_PALLAS) point_sum_host<PallasFp>(parts.data(), parts.size(), out);
        else point_sum_host<PallasFq>(parts.data(), parts.size(), out);
    }
    ~lurk_hip_msm_multi() {
        for (auto& sp : shards) {
            Shard* shp = sp.get();
            if (!shp->worker) continue;
            shp->worker->post([shp] {
                shp->ctx.reset();
                shp->staged.release();
            });
            try { shp->worker->wait(); } catch (...) {}
        }
    }
};

extern "C" {

int lurk_hip_msm_pallas(void* out, const void* bases, size_t n, const void* scalars, int is_mont) {
    return msm_oneshot(LURK_CURVE_PALLAS, out, bases, n, scalars, is_mont);
}
int lurk_hip_msm_vesta(void* out, const void* bases, size_t n, const void* scalars, int is_mont) {
    return msm_oneshot(LURK_CURVE_VESTA, out, bases, n, scalars, is_mont);
}

// pasta-msm's own C symbols: they return nothing (its CPU Pippenger cannot fail), so a failure here ends the process with the
// library's message - never a silent wrong commitment, never a CPU fallback
static void pasta_msm_symbol(int curve, void* out, const void* points, size_t npoints, const void* scalars, bool is_mont) {
    if (msm_oneshot(curve, out, points, npoints, scalars, is_mont ? 1 : 0) != 0) {
        fprintf(stderr, "liblurk_hip: mult_pippenger_%s failed: %s\n", curve == LURK_CURVE_PALLAS ? "pallas" : "vesta", lurk_hip_last_error());
        abort();
    }
}
void mult_pippenger_pallas(void* out, const void* points, size_t npoints, const void* scalars, bool is_mont) {
    pasta_msm_symbol(LURK_CURVE_PALLAS, out, points, npoints, scalars, is_mont);
}
void mult_pippenger_vesta(void* out, const void* points, size_t npoints, const void* scalars, bool is_mont) {
    pasta_msm_symbol(LURK_CURVE_VESTA, out, points, npoints, scalars, is_mont);
}

// pasta-msm's GPU entry points (its `cuda` feature, sppark's calling convention): the same arguments, a RustError {code, message} returned
// BY VALUE - message is a malloc'd C string the Rust side frees (sppark's `impl Drop for Error`), NULL on success.  What arecibo's GPU
// path binds instead of mult_pippenger_* (SURVEY.md section 8b).
static lurk_hip_rust_error rust_error_from(int rc) {
    lurk_hip_rust_error e;
    e.code = rc;
    e.message = rc == 0 ? nullptr : strdup(lurk_hip_last_error());
    return e;
}
lurk_hip_rust_error cuda_pippenger_pallas(void* out, const void* points, size_t npoints, const void* scalars, bool is_mont) {
    return rust_error_from(msm_oneshot(LURK_CURVE_PALLAS, out, points, npoints, scalars, is_mont ? 1 : 0));
}
lurk_hip_rust_error cuda_pippenger_vesta(void* out, const void* points, size_t npoints, const void* scalars, bool is_mont) {
    return rust_error_from(msm_oneshot(LURK_CURVE_VESTA, out, points, npoints, scalars, is_mont ? 1 : 0));
}

int lurk_hip_msm_oneshot_key_cache(int enable) {
    return guarded([&] { g_oneshot_key_cache.store(enable ? 1 : 0); });
}

int lurk_hip_msm_ctx_create(lurk_hip_msm_ctx** ctx, int curve, const void* bases, size_t n, int flags) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx pointer");
        LURK_REQUIRE(n == 0 || bases, "null bases");
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        DevBuf tmp(n * 64);
        if (n) LURK_HIP_CHECK(hipMemcpy(tmp.p, bases, n * 64, hipMemcpyHostToDevice));
        ctx_set_bases(c.get(), tmp.p, n, /*copy=*/true, flags, nullptr);
        *ctx = new lurk_hip_msm_ctx{std::move(c)};
    });
}
int lurk_hip_msm_ctx_create_dev(lurk_hip_msm_ctx** ctx, int curve, const void* d_bases, size_t n, int flags, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx pointer");
        LURK_REQUIRE(n == 0 || d_bases, "null bases");
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        ctx_set_bases(c.get(), d_bases, n, /*copy=*/false, flags, (hipStream_t)stream);
        *ctx = new lurk_hip_msm_ctx{std::move(c)};
    });
}
int lurk_hip_msm_ctx_run(lurk_hip_msm_ctx* ctx, void* out, const void* scalars, size_t n, int is_mont) {
    return guarded([&] {
        LURK_REQUIRE(ctx && out, "null argument");
        LURK_REQUIRE(n == 0 || scalars, "null scalars");
        DeviceGuard dg(ctx->impl->device);
        DevBuf ds(n * 32);
        if (n) LURK_HIP_CHECK(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
        ctx->impl->run(ds.p, n, is_mont, nullptr, out);
    });
}
int lurk_hip_msm_ctx_run_dev(lurk_hip_msm_ctx* ctx, void* out, const void* d_scalars, size_t n, int is_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(ctx && out, "null argument");
        LURK_REQUIRE(n == 0 || d_scalars, "null scalars");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->run(d_scalars, n, is_mont, (hipStream_t)stream, out);
    });
}
int lurk_hip_msm_ctx_submit_dev(lurk_hip_msm_ctx* ctx, int slot, const void* d_scalars, size_t n, int is_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx");
        LURK_REQUIRE(n == 0 || d_scalars, "null scalars");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->submit(slot, d_scalars, n, is_mont, (hipStream_t)stream, LURK_MSM_SUBMIT_DEFAULT);
    });
}
int lurk_hip_msm_ctx_submit_dev_mode(lurk_hip_msm_ctx* ctx, int slot, const void* d_scalars, size_t n, int is_mont, void* stream, int mode) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx");
        LURK_REQUIRE(n == 0 || d_scalars, "null scalars");
        LURK_REQUIRE(mode >= LURK_MSM_SUBMIT_DEFAULT && mode <= LURK_MSM_SUBMIT_FOLLOW, "unknown submit mode");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->submit(slot, d_scalars, n, is_mont, (hipStream_t)stream, mode);
    });
}
int lurk_hip_msm_ctx_submit_pair_dev(lurk_hip_msm_ctx* ctx, int slot, const void* d_scalars, size_t n, int is_mont, void* stream, int sel_bit) {
    return guarded([&] {
        LURK_REQUIRE(ctx && d_scalars, "null argument");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->submit_pair(slot, d_scalars, n, is_mont, (hipStream_t)stream, sel_bit);
    });
}
int lurk_hip_msm_ctx_wait_pair(lurk_hip_msm_ctx* ctx, int slot, void* out_lo, void* out_hi) {
    return guarded([&] {
        LURK_REQUIRE(ctx && out_lo && out_hi, "null argument");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->wait_pair(slot, out_lo, out_hi);
    });
}
int lurk_hip_msm_ctx_wait(lurk_hip_msm_ctx* ctx, int slot, void* out) {
    return guarded([&] {
        LURK_REQUIRE(ctx && out, "null argument");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->wait(slot, out);
    });
}
int lurk_hip_msm_ctx_destroy(lurk_hip_msm_ctx* ctx) {
    if (!ctx) return 0;
    return guarded([&] {
        DeviceGuard dg(ctx->impl->device);
        lurk::msm_ctx_drop_folded_child(ctx);
        delete ctx;
    });
}

// ---- key files ----------------------------------------------------------------------------------
// The reference keeps its public parameters - the commitment key is their bulk - in a disk cache and maps them back
// (/root/reference/src/public_parameters/mod.rs:33-56 "this clone is VERY expensive", disk_cache.rs:69-77).  A key file is the
// resident context's image: a 64-byte header, then the 64-byte affine records exactly as they sit in HBM (the bases; with
// with_table also the per-window multiples), so loading is open + mmap + copies straight into device memory, no parsing.
struct KeyFileHeader {
    char magic[8];  // "LURKHIPK"
    uint32_t version, curve, window_bits, windows;  // windows = 1: bases only
    uint64_t npoints, reserved[4];
};
static_assert(sizeof(KeyFileHeader) == 64, "key file header is 64 bytes");

int lurk_hip_msm_ctx_save(const lurk_hip_msm_ctx* ctx, const char* path, int with_table) {
    return guarded([&] {
        LURK_REQUIRE(ctx && path, "null argument");
        const MsmCtxBase& c = *ctx->impl;
        DeviceGuard dg(c.device);
        KeyFileHeader h{};
        memcpy(h.magic, "LURKHIPK", 8);
        h.version = 1;
        h.curve = (uint32_t)c.curve;
        h.window_bits = (uint32_t)c.c;
        h.windows = (with_table && c.precomputed && !c.small) ? (uint32_t)msm_num_windows(c.c) : 1u;  // the small form's table is rebuilt on load
        h.npoints = c.npoints;
        FILE* f = fopen(path, "wb");
        LURK_REQUIRE(f, std::string("cannot create ") + path);
        bool good = fwrite(&h, sizeof(h), 1, f) == 1;
        const size_t total = (size_t)h.windows * c.npoints * 64, chunk = (size_t)64 << 20;
        std::vector<char> buf(total < chunk ? total : chunk);
        for (size_t off = 0; good && off < total; off += chunk) {
            const size_t len = total - off < chunk ? total - off : chunk;
            if (hipMemcpy(buf.data(), (const char*)c.device_table() + off, len, hipMemcpyDeviceToHost) != hipSuccess) good = false;
            else good = fwrite(buf.data(), 1, len, f) == len;
        }
        good = (fclose(f) == 0) && good;
        LURK_REQUIRE(good, std::string("write failed: ") + path);
    });
}

int lurk_hip_msm_ctx_load(lurk_hip_msm_ctx** ctx, const char* path, int flags) {
    return guarded([&] {
        LURK_REQUIRE(ctx && path, "null argument");
        *ctx = nullptr;
        const int fd = open(path, O_RDONLY);
        LURK_REQUIRE(fd >= 0, std::string("cannot open ") + path);
        struct stat st;
        KeyFileHeader h{};
        bool good = fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(h) && pread(fd, &h, sizeof(h), 0) == (ssize_t)sizeof(h);
        good = good && memcmp(h.magic, "LURKHIPK", 8) == 0 && h.version == 1 && h.curve <= 1 && h.windows >= 1 && h.windows <= MSM_MAX_W &&
               (h.windows == 1 || (h.window_bits >= 16 && h.window_bits <= 20 && h.windows == (uint32_t)msm_num_windows((int)h.window_bits))) &&
               h.npoints < ((uint64_t)1 << 31) && (uint64_t)st.st_size == sizeof(h) + (uint64_t)h.windows * h.npoints * 64;
        if (!good) {
            close(fd);
            LURK_REQUIRE(false, std::string("not a lurk-hip key file (or truncated): ") + path);
        }
        const size_t n = h.npoints, total = (size_t)h.windows * n * 64;
        void* map = total ? mmap(nullptr, sizeof(h) + total, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        close(fd);
        LURK_REQUIRE(!total || map != MAP_FAILED, std::string("mmap failed: ") + path);
        std::unique_ptr<MsmCtxBase> c(new_ctx((int)h.curve));
        try {
            const bool want_table = (flags & LURK_MSM_FLAG_PRECOMPUTE) != 0;
            DevBuf buf(total);
            if (total) LURK_HIP_CHECK(hipMemcpy(buf.p, (const char*)map + sizeof(h), total, hipMemcpyHostToDevice));
            if (h.windows > 1 && want_table) {
                c->adopt_table(std::move(buf), n, true, (int)h.window_bits);  // the file's table as it is
            } else if (want_table) {
                ctx_set_bases(c.get(), buf.p, n, false, flags, nullptr);      // bases from the file, table rebuilt on the device
            } else {
                DevBuf bases(n * 64);  // bases only (drop a table the caller did not ask for)
                if (n) LURK_HIP_CHECK(hipMemcpy(bases.p, buf.p, n * 64, hipMemcpyDeviceToDevice));
                c->adopt_table(std::move(bases), n, false, MSM_C_PLAIN);
            }
        } catch (...) {
            if (map) munmap(map, sizeof(h) + total);
            throw;
        }
        if (map) munmap(map, sizeof(h) + total);
        *ctx = new lurk_hip_msm_ctx{std::move(c)};
    });
}
int lurk_hip_msm_ctx_rebind_dev(lurk_hip_msm_ctx* ctx, const void* d_bases, size_t npoints) {
    return guarded([&] {
        LURK_REQUIRE(ctx && (npoints == 0 || d_bases), "null argument");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->rebind(d_bases, npoints);
    });
}
int lurk_hip_msm_ctx_reserve(lurk_hip_msm_ctx* ctx, size_t nscalars, int slots) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx");
        DeviceGuard dg(ctx->impl->device);
        ctx->impl->reserve(nscalars, slots);
    });
}
int lurk_hip_msm_ctx_from_label(lurk_hip_msm_ctx** ctx, int curve, const void* label, size_t label_len, size_t npoints, int flags) {
    return guarded([&] {
        LURK_REQUIRE(ctx && (label || label_len == 0), "null argument");
        *ctx = nullptr;
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        DevBuf bases(npoints * 64);
        keygen_from_label_device(curve, label, label_len, npoints, bases.p, nullptr);
        if (flags & LURK_MSM_FLAG_PRECOMPUTE) ctx_set_bases(c.get(), bases.p, npoints, false, flags, nullptr);  // the table owns its copy
        else c->adopt_table(std::move(bases), npoints, false, MSM_C_PLAIN);
        *ctx = new lurk_hip_msm_ctx{std::move(c)};
    });
}
int lurk_hip_msm_ctx_device(const lurk_hip_msm_ctx* ctx, int* device) {
    return guarded([&] {
        LURK_REQUIRE(ctx && device, "null argument");
        *device = ctx->impl->device;
    });
}
int lurk_hip_msm_ctx_info(const lurk_hip_msm_ctx* ctx, int* curve, size_t* npoints, int* window_bits, int* precomputed) {
    return guarded([&] {
        LURK_REQUIRE(ctx, "null ctx");
        if (curve) *curve = ctx->impl->curve;
        if (npoints) *npoints = ctx->impl->npoints;
        if (window_bits) *window_bits = ctx->impl->c;
        if (precomputed) *precomputed = (ctx->impl->small || ctx->impl->precomputed) ? 1 : 0;  // a boolean, as before the small form existed
    });
}
int lurk_hip_msm_ctx_form(const lurk_hip_msm_ctx* ctx, int* form) {
    return guarded([&] {
        LURK_REQUIRE(ctx && form, "null argument");
        *form = ctx->impl->small ? LURK_MSM_FORM_SMALL : ctx->impl->precomputed ? LURK_MSM_FORM_TABLE : LURK_MSM_FORM_PLAIN;
    });
}

}  // extern "C"

namespace lurk {
// The folded key of the inner-product argument (ipa.hip) as a context that belongs to its parent key: created at the first proof, its
// points replaced (table rebuilt in place, workspaces kept) at every later one - creating and destroying a 65 536-point table key per
// proof cost 3-4 ms of hipMalloc / hipFree on the host.  One argument at a time holds it; a second one under the same key at the same
// moment gets a private context (owned = true).
struct FoldedChild {
    std::unique_ptr<lurk_hip_msm_ctx> ctx;
    std::mutex mu;
};
static std::mutex g_children_mu;
// (never destroyed: a key its owner forgot to destroy must not have its child's streams and buffers released by a static destructor
// after the HIP runtime has shut down)
static auto& g_children = *new std::map<const lurk_hip_msm_ctx*, std::unique_ptr<FoldedChild>>();

FoldedKeyLease::~FoldedKeyLease() {
    if (owned && ctx) (void)lurk_hip_msm_ctx_destroy(ctx);
}
FoldedKeyLease msm_ctx_folded_child(lurk_hip_msm_ctx* parent, const void* d_points, size_t m, hipStream_t s) {
    LURK_REQUIRE(parent && d_points && m, "null argument");
    const int curve = parent->impl->curve;
    const int flags = LURK_MSM_FLAG_PRECOMPUTE | LURK_MSM_FLAG_WINDOW_BITS(16);
    FoldedChild* fc = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_children_mu);
        auto& slot = g_children[parent];
        if (!slot) slot.reset(new FoldedChild);
        fc = slot.get();
    }
    FoldedKeyLease lease;
    lease.lk = std::unique_lock<std::mutex>(fc->mu, std::try_to_lock);
    if (!lease.lk.owns_lock()) {  // somebody else's argument holds the parent's child: a private one
        if (lurk_hip_msm_ctx_create_dev(&lease.ctx, curve, d_points, m, flags, (void*)s) != 0) throw HipFailure{LURK_HIP_ERR_HIP, lurk_hip_last_error()};
        lease.owned = true;
        return lease;
    }
    if (!fc->ctx) {
        std::unique_ptr<MsmCtxBase> c(new_ctx(curve));
        c->keep_buffers = true;
        ctx_set_bases(c.get(), d_points, m, /*copy=*/false, flags, s);
        fc->ctx.reset(new lurk_hip_msm_ctx{std::move(c)});
    } else {
        ctx_set_bases(fc->ctx->impl.get(), d_points, m, /*copy=*/false, flags, s);
    }
    lease.ctx = fc->ctx.get();
    return lease;
}
void msm_ctx_drop_folded_child(const lurk_hip_msm_ctx* parent) {
    std::unique_ptr<FoldedChild> dead;
    {
        std::lock_guard<std::mutex> lk(g_children_mu);
        auto it = g_children.find(parent);
        if (it == g_children.end()) return;
        dead = std::move(it->second);
        g_children.erase(it);
    }
    if (dead->ctx) {
        msm_ctx_drop_folded_child(dead->ctx.get());  // (a folded key long enough to have been folded again)
        dead->ctx.reset();
    }
}
void msm_ctx_wait_pair_xyzz(lurk_hip_msm_ctx* ctx, int slot, void* out_lo_xyzz128, void* out_hi_xyzz128) {
    LURK_REQUIRE(ctx && out_lo_xyzz128 && out_hi_xyzz128, "null argument");
    DeviceGuard dg(ctx->impl->device);
    ctx->impl->wait_pair_xyzz(slot, out_lo_xyzz128, out_hi_xyzz128);
}
MsmTableView msm_ctx_table_view(const lurk_hip_msm_ctx* ctx) {
    LURK_REQUIRE(ctx, "null ctx");
    const MsmCtxBase& c = *ctx->impl;
    MsmTableView v;
    v.table = c.device_table();
    v.npoints = c.npoints;
    v.curve = c.curve;
    v.window_bits = c.c;
    v.form = c.small ? LURK_MSM_FORM_SMALL : c.precomputed ? LURK_MSM_FORM_TABLE : LURK_MSM_FORM_PLAIN;
    v.windows = v.form == LURK_MSM_FORM_TABLE ? msm_num_windows(c.c) : 1;
    v.device = c.device;
    return v;
}
}  // namespace lurk

extern "C" {

// ---- one process, several devices --------------------------------------------------------------
int lurk_hip_msm_multi_create(lurk_hip_msm_multi** out, int curve, const void* bases, size_t n, const int* devices, int n_dev, int flags) {
    return guarded([&] {
        LURK_REQUIRE(out, "null ctx pointer");
        *out = nullptr;
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE(n == 0 || bases, "null bases");
        LURK_REQUIRE(devices && n_dev >= 1 && n_dev <= 64, "device list must hold 1..64 entries");
        const int have = lurk_hip_device_count();
        for (int i = 0; i < n_dev; i++) LURK_REQUIRE(devices[i] >= 0 && devices[i] < have, "device id out of range");
        auto m = std::make_unique<lurk_hip_msm_multi>();
        m->curve = curve;
        m->npoints = n;
        if (flags & LURK_MSM_FLAG_AUTO_SLICES) {
            // Every slice is a whole commitment: its own sort, plan and c - 1 reduction levels - ~0.64 ms of latency-bound chain on
            // one MI355X whatever its size - around 0.83 ms of accumulation per 2^20 points.  Below ~2^20 points
            // per slice the chain outweighs what another device takes off the accumulation (and on a list that repeats a device it is
            // pure overhead: +35 % for [0,0] at the rc = 100 step).  Use the first k devices of the list with k = max(1, n >> min_log).
            const char* e = getenv("LURK_MSM_MULTI_MIN_SLICE_LOG");
            int min_log = e ? atoi(e) : 20;
            if (min_log < 0) min_log = 0;
            if (min_log > 40) min_log = 40;
            size_t k = n >> min_log;
            if (k < 1) k = 1;
            if ((size_t)n_dev > k) n_dev = (int)k;
        }
        const size_t base = n / n_dev, extra = n % n_dev;  // the first n % n_dev shards hold one more point
        for (int i = 0; i < n_dev; i++) {
            auto sh = std::make_unique<lurk_hip_msm_multi::Shard>();
            sh->lo = (size_t)i * base + ((size_t)i < extra ? (size_t)i : extra);
            sh->hi = sh->lo + base + ((size_t)i < extra ? 1 : 0);
            sh->worker = std::make_unique<DeviceWorker>(devices[i]);
            m->shards.push_back(std::move(sh));
        }
        const char* hb = (const char*)bases;
        m->for_shards(n, [&](lurk_hip_msm_multi::Shard& sh, size_t cnt) {
            sh.ctx.reset(new_ctx(curve));
            DevBuf tmp(cnt * 64);
            LURK_HIP_CHECK(hipMemcpy(tmp.p, hb + sh.lo * 64, cnt * 64, hipMemcpyHostToDevice));
            ctx_set_bases(sh.ctx.get(), tmp.p, cnt, /*copy=*/true, flags, nullptr);
        });
        *out = m.release();
    });
}
int lurk_hip_msm_multi_shard(const lurk_hip_msm_multi* m, int index, int* device, size_t* first, size_t* count) {
    return guarded([&] {
        LURK_REQUIRE(m, "null ctx");
        LURK_REQUIRE(index >= 0 && (size_t)index < m->shards.size(), "shard index out of range");
        const auto& sh = *m->shards[index];
        if (device) *device = sh.worker->device();
        if (first) *first = sh.lo;
        if (count) *count = sh.hi - sh.lo;
    });
}
int lurk_hip_msm_multi_num_shards(const lurk_hip_msm_multi* m) { return m ? (int)m->shards.size() : 0; }

int lurk_hip_msm_multi_commit(lurk_hip_msm_multi* m, void* out, const void* scalars, size_t n, int is_mont) {
    return guarded([&] {
        LURK_REQUIRE(m && out, "null argument");
        LURK_REQUIRE(n <= m->npoints, "more scalars than bases in the context");
        LURK_REQUIRE(n == 0 || scalars, "null scalars");
        std::lock_guard<std::mutex> lk(m->mu);
        const char* hs = (const char*)scalars;
        m->for_shards(n, [&](lurk_hip_msm_multi::Shard& sh, size_t cnt) {
            sh.staged.ensure(cnt * 32);
            LURK_HIP_CHECK(hipMemcpy(sh.staged.p, hs + sh.lo * 32, cnt * 32, hipMemcpyHostToDevice));
            sh.ctx->run(sh.staged.p, cnt, is_mont, nullptr, &sh.partial);
        });
        m->sum(n, out);
    });
}
int lurk_hip_msm_multi_commit_dev(lurk_hip_msm_multi* m, void* out, const void* const* d_scalars, size_t n_slices, size_t n, int is_mont) {
    return guarded([&] {
        LURK_REQUIRE(m && out, "null argument");
        LURK_REQUIRE(n_slices == m->shards.size(), "one device pointer per shard is required (n_slices != number of shards)");
        LURK_REQUIRE(n <= m->npoints, "more scalars than bases in the context");
        LURK_REQUIRE(n == 0 || d_scalars, "null scalars");
        std::lock_guard<std::mutex> lk(m->mu);
        for (size_t i = 0; i < m->shards.size(); i++)
            LURK_REQUIRE(m->shards[i]->lo >= n || m->shards[i]->lo == m->shards[i]->hi || d_scalars[i], "null shard pointer");
        m->for_shards(n, [&](lurk_hip_msm_multi::Shard& sh, size_t cnt) {
            size_t idx = 0;
            while (m->shards[idx].get() != &sh) idx++;
            sh.ctx->run(d_scalars[idx], cnt, is_mont, nullptr, &sh.partial);
        });
        m->sum(n, out);
    });
}
// Asynchronous form: the slices of one commitment are submitted on slot `slot` of every slice's context from the calling thread
// (each under its own device guard: a submit only enqueues) and run concurrently on their devices; wait collects the partial
// commitments in slice order and sums them with the host group law.  after_streams[i] (may be NULL) is the stream ON SLICE i's
// DEVICE that produced slice i's scalars, e.g. the stream a peer copy into that device was enqueued on.
int lurk_hip_msm_multi_submit_dev(lurk_hip_msm_multi* m, int slot, const void* const* d_scalars, void* const* after_streams, size_t n_slices, size_t n,
                                  int is_mont, int mode) {
    return guarded([&] {
        LURK_REQUIRE(m, "null ctx");
        LURK_REQUIRE(slot >= 0 && slot < MSM_SLOTS, "slot out of range");
        LURK_REQUIRE(n_slices == m->shards.size(), "one device pointer per shard is required (n_slices != number of shards)");
        LURK_REQUIRE(n <= m->npoints, "more scalars than bases in the context");
        LURK_REQUIRE(n == 0 || d_scalars, "null scalars");
        LURK_REQUIRE(mode >= LURK_MSM_SUBMIT_DEFAULT && mode <= LURK_MSM_SUBMIT_FOLLOW, "unknown submit mode");
        std::lock_guard<std::mutex> lk(m->mu);
        LURK_REQUIRE(!m->pending[slot], "slot is busy: wait for it first");
        for (size_t i = 0; i < m->shards.size(); i++)
            LURK_REQUIRE(m->shards[i]->lo >= n || m->shards[i]->lo == m->shards[i]->hi || d_scalars[i], "null shard pointer");
        size_t done = 0;
        try {
            for (; done < m->shards.size(); done++) {
                auto& sh = *m->shards[done];
                if (sh.lo >= n || sh.lo == sh.hi) continue;
                const size_t cnt = (sh.hi < n ? sh.hi : n) - sh.lo;
                DeviceGuard dg(sh.worker->device());
                sh.ctx->submit(slot, d_scalars[done], cnt, is_mont, after_streams ? (hipStream_t)after_streams[done] : nullptr, mode);
            }
        } catch (...) {  // what was submitted is drained: the key stays usable
            for (size_t i = 0; i < done; i++) {
                auto& sh = *m->shards[i];
                if (sh.lo >= n || sh.lo == sh.hi) continue;
                try {
                    DeviceGuard dg(sh.worker->device());
                    sh.ctx->wait(slot, &sh.partial);
                } catch (...) {
                }
            }
            throw;
        }
        m->pending[slot] = true;
        m->pending_n[slot] = n;
    });
}
int lurk_hip_msm_multi_wait(lurk_hip_msm_multi* m, int slot, void* out) {
    return guarded([&] {
        LURK_REQUIRE(m && out, "null argument");
        LURK_REQUIRE(slot >= 0 && slot < MSM_SLOTS, "slot out of range");
        std::lock_guard<std::mutex> lk(m->mu);
        LURK_REQUIRE(m->pending[slot], "nothing was submitted on this slot");
        m->pending[slot] = false;
        const size_t n = m->pending_n[slot];
        std::unique_ptr<HipFailure> first;
        std::vector<Jacobian<PallasFp>> parts;
        for (auto& sp : m->shards) {
            auto& sh = *sp;
            if (sh.lo >= n || sh.lo == sh.hi) continue;
            Jacobian<PallasFp> part;
            try {  // every slice is waited for even if one fails: nothing stays in flight
                DeviceGuard dg(sh.worker->device());
                sh.ctx->wait(slot, &part);
                parts.push_back(part);
            } catch (const HipFailure& e) {
                if (!first) first.reset(new HipFailure(e));
            }
        }
        if (first) throw *first;
        if (m->curve == LURK_CURVE_PALLAS) point_sum_host<PallasFp>(parts.data(), parts.size(), out);
        else point_sum_host<PallasFq>(parts.data(), parts.size(), out);
    });
}
int lurk_hip_msm_multi_destroy(lurk_hip_msm_multi* m) {
    if (!m) return 0;
    return guarded([&] { delete m; });
}

// host-side group helpers (a handful of points: partial commitments gathered from the ranks)
int lurk_hip_point_sum_gathered(int curve, void* out, const void* gathered, size_t world) {
    if (world == 0) {
        set_error(LURK_HIP_ERR_INVALID_ARG, "lurk_hip_point_sum_gathered: a world of 0 ranks");
        return LURK_HIP_ERR_INVALID_ARG;
    }
    return lurk_hip_point_sum(curve, out, gathered, world);
}

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
int lurk_hip_point_mul(int curve, void* out, const void* point, const void* scalar32, int is_mont) {
    try {
        LURK_REQUIRE(curve == 0 || curve == 1, "unknown curve id");
        LURK_REQUIRE(out && point && scalar32, "null argument");
        if (curve == 0) point_mul_host<PallasFp, PallasFq>(point, scalar32, is_mont, out);
        else point_mul_host<PallasFq, PallasFp>(point, scalar32, is_mont, out);
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

// common.hpp - host runtime glue shared by the .hip translation units of liblurk_hip.so:
// error reporting behind the C ABI, device selection, RAII device memory, per-kernel HIP-event
// timing on the launch stream.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/lurk_hip.h"

namespace lurk {

// thread-local last error (lurk_hip_last_error)
void set_error(int code, const std::string& msg);
int last_error_code();

struct HipFailure {
    int code;
    std::string msg;
};

#define LURK_HIP_CHECK(expr)                                                                              \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            (void)hipGetLastError(); /* the failure becomes an exception: the thread's sticky last-error must not resurface in a later hipGetLastError() check */ \
            throw ::lurk::HipFailure{_e == hipErrorOutOfMemory ? LURK_HIP_ERR_OOM : LURK_HIP_ERR_HIP,     \
                                     std::string(#expr) + ": " + hipGetErrorString(_e)};                 \
        }                                                                                                 \
    } while (0)

#define LURK_REQUIRE(cond, msg)                                                       \
    do {                                                                              \
        if (!(cond)) throw ::lurk::HipFailure{LURK_HIP_ERR_INVALID_ARG, std::string(msg)}; \
    } while (0)

// Fails loudly (no CPU fallback) unless a gfx950 device is usable.
void require_device();

// Every handle (MSM context, R1CS shape, NTT plan) records the device it was created on; its entry points run
// under a DeviceGuard, so a handle may be used from any host thread whatever that thread's current device is
// (hipSetDevice is per thread).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        LURK_HIP_CHECK(hipGetDevice(&prev));
        if (prev != dev) {
            LURK_HIP_CHECK(hipSetDevice(dev));
            switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
int current_device();

// More than 64 KB of dynamic LDS has to be requested per kernel and per device; done once for each pair.
void allow_dynamic_lds(const void* kernel, int bytes);

// Wraps the body of a C-ABI entry point: converts exceptions to error codes.
template <class F>
int guarded(F&& f) {
    try {
        require_device();
        f();
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    } catch (const std::exception& e) {
        set_error(LURK_HIP_ERR_HIP, e.what());
        return LURK_HIP_ERR_HIP;
    }
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        LURK_HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    void ensure(size_t n) {
        if (n > bytes) alloc(n);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- per-kernel timing -------------------------------------------------------------------
// When enabled, launches go through Profiler::Scope which records a HIP event pair on the launch
// stream around the kernel; totals are resolved lazily (events are synchronised on query).
class Profiler {
  public:
    static Profiler& get();
    void enable(bool on) { enabled_ = on; }
    bool enabled() const { return enabled_; }
    void reset();
    hipEvent_t begin(hipStream_t s);                          // records and returns the opening event
    void end(const char* name, hipEvent_t a, hipStream_t s);  // records the closing event
    void query(const char* prefix, double* total_ms, uint64_t* launches);

  private:
    struct Rec { std::string name; hipEvent_t a, b; int device; hipStream_t s; };
    std::mutex mu_;
    bool enabled_ = false;
    std::vector<Rec> recs_;
    std::map<int, std::vector<hipEvent_t>> pool_;  // events belong to the device they were created on
    std::map<std::string, std::pair<double, uint64_t>> done_;
    hipEvent_t get_event(int device);
    void dump_timeline();  // (mu_ held)
};

// scopes may be open on several host threads / devices at once (the multi-device MSM): the opening event travels
// with the scope
// set while the calling thread records launches into a hipGraph (msm.hip): timing events do not belong in the recording
inline thread_local bool tl_capturing = false;

struct ProfScope {
    hipStream_t s;
    const char* name;
    hipEvent_t a = nullptr;
    ProfScope(const char* nm, hipStream_t st) : s(st), name(nm) {
        if (Profiler::get().enabled() && !tl_capturing) a = Profiler::get().begin(s);
    }
    ~ProfScope() {
        if (a) Profiler::get().end(name, a, s);
    }
};

// One host thread bound to one device: the multi-device entry points post their per-device work here so that
// the devices are driven concurrently from a single-process caller (arecibo's prover is one process,
// /root/reference/src/proof/nova.rs:304-326).
class DeviceWorker {
  public:
    explicit DeviceWorker(int device);
    ~DeviceWorker();
    void post(std::function<void()> job);  // one job at a time: post, then wait
    void wait();                           // blocks until the job is done; rethrows its HipFailure
    int device() const { return device_; }

  private:
    void loop();
    int device_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool has_job_ = false, busy_ = false, stop_ = false, failed_ = false;
    HipFailure err_{0, ""};
    std::thread th_;
};

// A commitment key's device-resident points as the other translation units may read them (msm.hip): table[w * npoints + i] =
// 2^(window_bits w) P_i, 64-byte Montgomery affine records ((0, 0) = identity); windows = 1 for a plain key (and for the small-commitment
// form, whose multiples table is not exposed: its first npoints records are the points themselves)
struct MsmTableView {
    const void* table = nullptr;
    size_t npoints = 0;
    int curve = 0, window_bits = 0, windows = 1, form = 0, device = 0;
};
MsmTableView msm_ctx_table_view(const lurk_hip_msm_ctx* ctx);
// lurk_hip_msm_ctx_wait_pair without the normalisation: the two commitments as XYZZ points (4 x 32 B each, Montgomery limbs) - the caller
// adds to them and normalises both with one field inversion (ipa.hip: ~17 us of host time per inversion, five per round before)
void msm_ctx_wait_pair_xyzz(lurk_hip_msm_ctx* ctx, int slot, void* out_lo_xyzz128, void* out_hi_xyzz128);
// the parent key's folded-key context with its points replaced by d_points (m affine records on the device): msm.hip
struct FoldedKeyLease {
    lurk_hip_msm_ctx* ctx = nullptr;
    bool owned = false;  // a private context, destroyed with the lease
    std::unique_lock<std::mutex> lk;
    FoldedKeyLease() = default;
    FoldedKeyLease(FoldedKeyLease&& o) noexcept : ctx(o.ctx), owned(o.owned), lk(std::move(o.lk)) {
        o.ctx = nullptr;
        o.owned = false;
    }
    FoldedKeyLease(const FoldedKeyLease&) = delete;
    ~FoldedKeyLease();
};
FoldedKeyLease msm_ctx_folded_child(lurk_hip_msm_ctx* parent, const void* d_points, size_t m, hipStream_t s);
void msm_ctx_drop_folded_child(const lurk_hip_msm_ctx* parent);

// Prover scratch: ONE stack of device memory per (device, stream), kept for the life of the process.  The compressing prover takes ~60
// scratch vectors per proof (sum-check tables, eq tables, the opening argument's halves); as hipMallocAsync / hipFreeAsync pairs they
// cost the library prover ~2 ms per 2^20-row proof over a caller with a caching allocator (round 4: 34.3 against 32.1 ms).  An
// ArenaBuf is an automatic variable: construction pushes (growing the arena by whole blocks the first time a size is seen),
// destruction pops; every user of one stream's arena is ordered on that stream, and a thread holds the arena (recursive lock) for as
// long as it has a buffer alive, so two provers never share it.
class ScratchArena {
  public:
    static ScratchArena& of(hipStream_t s);  // the arena of (current device, s)
    static size_t trim_all();                // lurk_hip_scratch_trim: hipFree every block of every arena with nothing live; bytes released
    void* push(size_t bytes);
    void pop(void* p);
    std::recursive_mutex mu;

  private:
    struct Block { char* base; size_t cap, top; };
    struct Live { void* p; size_t block, prev_top; bool freed; };
    std::vector<Block> blocks_;
    std::vector<Live> live_;
    size_t cur_ = 0;
};
struct ArenaBuf {
    ScratchArena* arena;
    void* p = nullptr;
    ArenaBuf(size_t bytes, hipStream_t s) : arena(&ScratchArena::of(s)) {
        arena->mu.lock();
        try {
            p = arena->push(bytes ? bytes : 32);
        } catch (...) {
            arena->mu.unlock();
            throw;
        }
    }
    ~ArenaBuf() {
        arena->pop(p);
        arena->mu.unlock();
    }
    ArenaBuf(const ArenaBuf&) = delete;
    ArenaBuf& operator=(const ArenaBuf&) = delete;
};

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

int num_cus();  // multiprocessor count of the current device
// The current device's stream-ordered memory pool keeps up to 4 GiB of freed blocks instead of handing them back to the driver at the
// next synchronisation (the default release threshold is 0): the provers take their scratch from it (hipMallocAsync) dozens of times per
// proof with synchronisations in between - a proof of a 2^20-row instance spent ~5 ms re-allocating.  Once per device, cheap afterwards.
void stream_pool_retain();

}  // namespace lurk

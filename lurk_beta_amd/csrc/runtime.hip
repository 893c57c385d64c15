// runtime.hip - error state, device checks, profiler, misc C-ABI entry points.
#include "common.hpp"

namespace lurk {

static thread_local int tl_code = 0;
static thread_local std::string tl_msg;

void set_error(int code, const std::string& msg) {
    tl_code = code;
    tl_msg = msg;
}
int last_error_code() { return tl_code; }

static int g_device_state = -1;  // -1 unknown, 0 none, 1 ok
static std::string g_device_msg;
static std::mutex g_device_mu;
static int g_num_cus = 0;

void require_device() {
    std::lock_guard<std::mutex> lk(g_device_mu);
    if (g_device_state < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0) {
            g_device_state = 0;
            g_device_msg = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                           "); liblurk_hip has no CPU fallback";
        } else {
            int dev = 0;
            (void)hipGetDevice(&dev);
            hipDeviceProp_t prop;
            e = hipGetDeviceProperties(&prop, dev);
            if (e != hipSuccess) {
                g_device_state = 0;
                g_device_msg = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
            } else if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
                g_device_state = 0;
                g_device_msg = std::string("device is ") + prop.gcnArchName + ", liblurk_hip is built for gfx950 only";
            } else {
                g_device_state = 1;
                g_num_cus = prop.multiProcessorCount;
            }
        }
    }
    if (g_device_state == 0) throw HipFailure{LURK_HIP_ERR_NO_DEVICE, g_device_msg};
}

int num_cus() { return g_num_cus > 0 ? g_num_cus : 256; }

int current_device() {
    int dev = 0;
    LURK_HIP_CHECK(hipGetDevice(&dev));
    return dev;
}

void allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;  // (kernel, device) -> bytes granted
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(mu);
    int& have = done[std::make_pair(kernel, dev)];
    if (have >= bytes) return;
    LURK_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
}

void stream_pool_retain() {
    static std::mutex mu;
    static bool done[64] = {false};
    const int dev = current_device();
    if (dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return;
    done[dev] = true;
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess || !pool) return;
    uint64_t keep = (uint64_t)4 << 30;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
}

Profiler& Profiler::get() {
    static Profiler p;
    return p;
}
hipEvent_t Profiler::get_event(int device) {
    auto& pool = pool_[device];
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    LURK_HIP_CHECK(hipEventCreate(&e));
    return e;
}
hipEvent_t Profiler::begin(hipStream_t s) {
    const int dev = current_device();
    hipEvent_t a;
    {
        std::lock_guard<std::mutex> lk(mu_);
        a = get_event(dev);
    }
    LURK_HIP_CHECK(hipEventRecord(a, s));
    return a;
}
void Profiler::end(const char* name, hipEvent_t a, hipStream_t s) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(mu_);
    hipEvent_t b = get_event(dev);
    (void)hipEventRecord(b, s);  // called from a destructor: never throws
    recs_.push_back({name, a, b, dev, s});
}
// LURK_PROF_TIMELINE=<file>: the scopes recorded since the last query / reset, as a device-side timeline (start relative to the first
// scope's opening event, microseconds; one line per scope: start, end, duration, stream, name) - what rocprofv3 cannot give for commitments
// in flight, since under it every launch call costs the submitting thread ~30 us and the queues drift apart (bench_tools/scope_timeline.py)
void Profiler::dump_timeline() {
    static const char* tl = getenv("LURK_PROF_TIMELINE");
    if (!tl || recs_.empty()) return;
    FILE* f = fopen(tl, "a");
    if (!f) return;
    fprintf(f, "# %zu scopes\n", recs_.size());
    for (auto& r : recs_) {
        if (r.device != recs_[0].device) continue;
        DeviceGuard g(r.device);
        float t0 = 0, t1 = 0;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t0, recs_[0].a, r.a) != hipSuccess ||
            hipEventElapsedTime(&t1, recs_[0].a, r.b) != hipSuccess)
            continue;
        fprintf(f, "%10.1f %10.1f %8.1f  %p  %s\n", t0 * 1e3, t1 * 1e3, (t1 - t0) * 1e3, (void*)r.s, r.name.c_str());
    }
    fclose(f);
}
void Profiler::reset() {
    std::lock_guard<std::mutex> lk(mu_);
    dump_timeline();
    for (auto& r : recs_) {
        DeviceGuard g(r.device);
        (void)hipEventSynchronize(r.b);
        pool_[r.device].push_back(r.a);
        pool_[r.device].push_back(r.b);
    }
    recs_.clear();
    done_.clear();
}
void Profiler::query(const char* prefix, double* total_ms, uint64_t* launches) {
    std::lock_guard<std::mutex> lk(mu_);
    dump_timeline();
    for (auto& r : recs_) {
        DeviceGuard g(r.device);
        LURK_HIP_CHECK(hipEventSynchronize(r.b));
        float ms = 0;
        LURK_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        auto& d = done_[r.name];
        d.first += ms;
        d.second += 1;
        pool_[r.device].push_back(r.a);
        pool_[r.device].push_back(r.b);
    }
    recs_.clear();
    double tot = 0;
    uint64_t cnt = 0;
    size_t pl = strlen(prefix);
    for (auto& kv : done_)
        if (kv.first.compare(0, pl, prefix) == 0) {
            tot += kv.second.first;
            cnt += kv.second.second;
        }
    *total_ms = tot;
    *launches = cnt;
}

static std::mutex g_arena_map_mu;
static std::map<std::pair<int, hipStream_t>, ScratchArena*>& arena_map() {
    static auto& arenas = *new std::map<std::pair<int, hipStream_t>, ScratchArena*>();  // never torn down (process lifetime)
    return arenas;
}
ScratchArena& ScratchArena::of(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_arena_map_mu);
    auto& arenas = arena_map();
    auto key = std::make_pair(current_device(), s);
    auto it = arenas.find(key);
    if (it == arenas.end()) it = arenas.emplace(key, new ScratchArena()).first;
    return *it->second;
}
// An arena that some thread holds (a prover with a buffer alive keeps its recursive lock) is skipped: try_lock, never wait.
size_t ScratchArena::trim_all() {
    std::lock_guard<std::mutex> g(g_arena_map_mu);
    size_t released = 0;
    for (auto& kv : arena_map()) {
        ScratchArena& a = *kv.second;
        if (!a.mu.try_lock()) continue;
        if (a.live_.empty()) {
            DeviceGuard dg(kv.first.first);
            for (auto& b : a.blocks_) {
                (void)hipFree(b.base);
                released += b.cap;
            }
            a.blocks_.clear();
            a.cur_ = 0;
        }
        a.mu.unlock();
    }
    return released;
}
void* ScratchArena::push(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    // the stack runs over a chain of blocks: the current one, else the next one if it is empty and large enough, else a new block
    // spliced in behind the current one (a repeated sequence of pushes and pops finds every block where the first run left it)
    if (blocks_.empty() || blocks_[cur_].cap - blocks_[cur_].top < bytes) {
        size_t next = blocks_.empty() ? 0 : cur_ + 1;
        if (next >= blocks_.size() || blocks_[next].top != 0 || blocks_[next].cap < bytes) {
            const size_t cap = bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);
            void* base = nullptr;
            LURK_HIP_CHECK(hipMalloc(&base, cap));
            blocks_.insert(blocks_.begin() + next, Block{(char*)base, cap, 0});
            for (auto& l : live_)
                if (l.block >= next) l.block++;
        }
        cur_ = next;
    }
    Block& b = blocks_[cur_];
    void* p = b.base + b.top;
    live_.push_back(Live{p, cur_, b.top, false});
    b.top += bytes;
    return p;
}
void ScratchArena::pop(void* p) {
    for (size_t i = live_.size(); i-- > 0;)
        if (live_[i].p == p && !live_[i].freed) {
            live_[i].freed = true;
            break;
        }
    while (!live_.empty() && live_.back().freed) {  // automatic variables die in reverse order: this loop runs once per pop
        blocks_[live_.back().block].top = live_.back().prev_top;
        cur_ = live_.back().block;
        live_.pop_back();
    }
    if (live_.empty()) cur_ = 0;
}

DeviceWorker::DeviceWorker(int device) : device_(device), th_([this] { loop(); }) {}
DeviceWorker::~DeviceWorker() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
}
void DeviceWorker::loop() {
    bool bound = hipSetDevice(device_) == hipSuccess;
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return has_job_ || stop_; });
            if (stop_ && !has_job_) return;
            job = std::move(job_);
            has_job_ = false;
        }
        bool failed = false;
        HipFailure err{0, ""};
        try {
            if (!bound) throw HipFailure{LURK_HIP_ERR_HIP, "hipSetDevice(" + std::to_string(device_) + ") failed on the worker thread"};
            job();
        } catch (const HipFailure& e) {
            failed = true;
            err = e;
        } catch (const std::exception& e) {
            failed = true;
            err = HipFailure{LURK_HIP_ERR_HIP, e.what()};
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            failed_ = failed;
            err_ = err;
            busy_ = false;
        }
        cv_.notify_all();
    }
}
void DeviceWorker::post(std::function<void()> job) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return !busy_; });
    job_ = std::move(job);
    has_job_ = true;
    busy_ = true;
    failed_ = false;
    lk.unlock();
    cv_.notify_all();
}
void DeviceWorker::wait() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return !busy_; });
    if (failed_) {
        failed_ = false;
        throw err_;
    }
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char* lurk_hip_last_error(void) {
    static thread_local std::string copy;
    if (lurk::last_error_code() == 0) return "";
    copy = std::string("lurk_hip error ") + std::to_string(lurk::last_error_code()) + ": " + lurk::tl_msg;
    return copy.c_str();
}
const char* lurk_hip_version(void) { return "lurk-hip 0.2 (gfx950)"; }
int lurk_hip_abi_version(void) { return LURK_HIP_ABI_VERSION; }

int lurk_hip_scratch_trim(size_t* released_bytes) {
    return guarded([&] {
        const size_t n = ScratchArena::trim_all();
        if (released_bytes) *released_bytes = n;
    });
}

int lurk_hip_set_device(int device) {
    return guarded([&] {
        LURK_HIP_CHECK(hipSetDevice(device));
    });
}
int lurk_hip_profile_enable(int on) {
    Profiler::get().enable(on != 0);
    return 0;
}
int lurk_hip_profile_reset(void) {
    return guarded([&] { Profiler::get().reset(); });
}
int lurk_hip_profile_get(const char* prefix, double* total_ms, uint64_t* launches) {
    return guarded([&] {
        LURK_REQUIRE(prefix && total_ms && launches, "null argument");
        Profiler::get().query(prefix, total_ms, launches);
    });
}
}

// queue_hop_probe.hip - how long does a kernel on stream B wait after the event it depends on (recorded behind a kernel on stream A) has
// fired?  With and without a long compute-bound kernel occupying the device, for B at normal / high stream priority, and for the same
// two kernels on ONE stream (a plain kernel boundary).  Device timestamps (wall_clock64, 100 MHz) written by the kernels themselves.
// Why: the folding step's commit(T) chain crosses two queues (copies -> cross term -> T's slot) and T's first sort kernel was seen to start
// ~0.6 ms after the cross term had finished.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void stamp_end(uint64_t* t, int spin) {  // a short kernel: spins a little, stamps when it ends
    uint64_t s = wall_clock64();
    while (wall_clock64() - s < (uint64_t)spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = wall_clock64();
}
__global__ void stamp_start(uint64_t* t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ __launch_bounds__(256) void hog(uint32_t* out, int iters) {  // VALU-bound, fills every SIMD (many waves)
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9E3779B9u;
    for (int i = 0; i < iters; i++) {
        a = a * 1664525u + b;
        b = b * 22695477u + a;
    }
    if (a == 0x12345 && b == 0x6789) out[0] = a;
}

static double run_case(bool cross_stream, bool high_prio, bool loaded, int hog_blocks, int hog_iters) {
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sa, sb, sh;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, high_prio ? greatest : 0));
    CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint64_t* d_t;
    uint32_t* d_o;
    CK(hipMalloc(&d_t, 16));
    CK(hipMalloc(&d_o, 16));
    std::vector<double> gaps;
    for (int rep = 0; rep < 12; rep++) {
        CK(hipMemset(d_t, 0, 16));
        CK(hipDeviceSynchronize());
        if (loaded) hipLaunchKernelGGL(hog, dim3(hog_blocks), dim3(256), 0, sh, d_o, hog_iters);
        hipLaunchKernelGGL(stamp_end, dim3(1), dim3(64), 0, sa, d_t, 20000);  // 200 us: the hog is well under way when it ends
        if (cross_stream) {
            CK(hipEventRecord(ev, sa));
            CK(hipStreamWaitEvent(sb, ev, 0));
            hipLaunchKernelGGL(stamp_start, dim3(1), dim3(64), 0, sb, d_t);
        } else {
            hipLaunchKernelGGL(stamp_start, dim3(1), dim3(64), 0, sa, d_t);
        }
        CK(hipDeviceSynchronize());
        uint64_t h[2];
        CK(hipMemcpy(h, d_t, 16, hipMemcpyDeviceToHost));
        if (rep >= 2) gaps.push_back((double)(int64_t)(h[1] - h[0]) / 100.0);  // 100 MHz -> us
    }
    double sum = 0, mx = 0;
    for (double g : gaps) { sum += g; mx = g > mx ? g : mx; }
    CK(hipFree(d_t));
    CK(hipFree(d_o));
    CK(hipEventDestroy(ev));
    CK(hipStreamDestroy(sa));
    CK(hipStreamDestroy(sb));
    CK(hipStreamDestroy(sh));
    printf("{\"cross_stream\": %d, \"consumer_high_priority\": %d, \"device_loaded\": %d, \"gap_us_mean\": %.1f, \"gap_us_max\": %.1f}\n", (int)cross_stream,
           (int)high_prio, (int)loaded, sum / gaps.size(), mx);
    return sum / gaps.size();
}

int main() {
    // hog: 256 CUs x 12 blocks of 256 threads x enough iterations for ~2 ms
    const int hog_blocks = 256 * 12, hog_iters = 400000;
    for (int loaded = 0; loaded <= 1; loaded++) {
        run_case(false, false, loaded, hog_blocks, hog_iters);
        run_case(true, false, loaded, hog_blocks, hog_iters);
        run_case(true, true, loaded, hog_blocks, hog_iters);
    }
    return 0;
}

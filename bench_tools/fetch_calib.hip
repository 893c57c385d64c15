// fetch_calib.hip - calibrates rocprofv3's FETCH_SIZE on gfx950 for the two access patterns the MSM kernels have, as
// MI355X_MICROARCH.md (section HBM) asks: "Other access widths ... are uncalibrated: calibrate on a known byte count in your own
// access pattern before trusting an absolute."  Two kernels with a KNOWN byte count each:
//   calib_stream    every lane reads 16 B of a contiguous buffer (the pattern the guide calibrated: FETCH_SIZE reports 1/2)
//   calib_gather64  every lane reads one 64-byte record (4 x 16 B, as `table[e]` of msm_accumulate_kernel compiles) at a pseudo-random
//                   index of a buffer far larger than the 256 MiB Infinity Cache
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`: bench_tools/fetch_calib.sh divides the reported KiB by the known bytes.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void calib_stream(const uint4* __restrict__ in, size_t n16, uint4* __restrict__ out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = in[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;  // keeps the loads alive
}

__global__ __launch_bounds__(256) void calib_gather64(const uint4* __restrict__ table, size_t records, size_t gathers, uint4* __restrict__ out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < gathers; i += (size_t)gridDim.x * 256) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;  // splitmix-style index: no two neighbouring lanes share a line
        z ^= z >> 31;
        z *= 0x94D049BB133111EBull;
        z ^= z >> 29;
        const uint4* r = table + (z % records) * 4;
        const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc.x == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t table_bytes = (size_t)13 << 28;  // 3.25 GiB: the table of a 2^22-point key
    const size_t gathers = (size_t)13 << 22;      // 54.5 M records of 64 B: one 2^22-point commitment's worth
    const size_t stream_bytes = (size_t)2 << 30;
    uint4 *table = nullptr, *out = nullptr;
    CK(hipMalloc(&table, table_bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(table, 1, table_bytes));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_stream, dim3(256 * 8), dim3(256), 0, nullptr, table, stream_bytes / 16, out);
        hipLaunchKernelGGL(calib_gather64, dim3(256 * 8), dim3(256), 0, nullptr, table, table_bytes / 64, gathers, out);
    }
    CK(hipDeviceSynchronize());
    printf("{\"calib_stream_bytes\": %zu, \"calib_gather64_bytes\": %zu, \"gather_records\": %zu, \"table_bytes\": %zu}\n", stream_bytes, gathers * 64, gathers,
           table_bytes);
    return 0;
}

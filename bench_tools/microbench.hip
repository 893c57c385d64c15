// microbench.hip - instruction-rate and field-multiplier probes for gfx950 (tooling, not product).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 bench_tools/microbench.hip -o bench_tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define LURK_MUL_FORCE_INLINE_OFF
#include "../lurk_beta_amd/csrc/field.cuh"
#include "../lurk_beta_amd/csrc/field29.cuh"
using namespace lurk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

__global__ void k_mad64(uint32_t* out, uint32_t seed) {
    uint64_t a0 = (uint64_t)(seed + threadIdx.x + 0); uint64_t a1 = (uint64_t)(seed + threadIdx.x + 1); uint64_t a2 = (uint64_t)(seed + threadIdx.x + 2); uint64_t a3 = (uint64_t)(seed + threadIdx.x + 3); uint64_t a4 = (uint64_t)(seed + threadIdx.x + 4); uint64_t a5 = (uint64_t)(seed + threadIdx.x + 5); uint64_t a6 = (uint64_t)(seed + threadIdx.x + 6); uint64_t a7 = (uint64_t)(seed + threadIdx.x + 7);
    uint32_t x = seed * 3 + threadIdx.x, y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_mul_lo(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_mul_hi(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_mul_u24(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_mul_u32_u24_e32 %0, %0, %8\n v_mul_u32_u24_e32 %1, %1, %8\n v_mul_u32_u24_e32 %2, %2, %8\n v_mul_u32_u24_e32 %3, %3, %8\n v_mul_u32_u24_e32 %4, %4, %8\n v_mul_u32_u24_e32 %5, %5, %8\n v_mul_u32_u24_e32 %6, %6, %8\n v_mul_u32_u24_e32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_add_u32(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_add_u32_e32 %0, %0, %8\n v_add_u32_e32 %1, %1, %8\n v_add_u32_e32 %2, %2, %8\n v_add_u32_e32 %3, %3, %8\n v_add_u32_e32 %4, %4, %8\n v_add_u32_e32 %5, %5, %8\n v_add_u32_e32 %6, %6, %8\n v_add_u32_e32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_xor(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_xor_b32_e32 %0, %0, %8\n v_xor_b32_e32 %1, %1, %8\n v_xor_b32_e32 %2, %2, %8\n v_xor_b32_e32 %3, %3, %8\n v_xor_b32_e32 %4, %4, %8\n v_xor_b32_e32 %5, %5, %8\n v_xor_b32_e32 %6, %6, %8\n v_xor_b32_e32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_fma64(uint32_t* out, uint32_t seed) {
    double a0 = (double)(seed + threadIdx.x + 0); double a1 = (double)(seed + threadIdx.x + 1); double a2 = (double)(seed + threadIdx.x + 2); double a3 = (double)(seed + threadIdx.x + 3); double a4 = (double)(seed + threadIdx.x + 4); double a5 = (double)(seed + threadIdx.x + 5); double a6 = (double)(seed + threadIdx.x + 6); double a7 = (double)(seed + threadIdx.x + 7);
    double x = 1.0000001, y = 1e-9;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_lshr64(uint32_t* out, uint32_t seed) {
    uint64_t a0 = (uint64_t)(seed + threadIdx.x + 0) << 40; uint64_t a1 = (uint64_t)(seed + threadIdx.x + 1) << 40; uint64_t a2 = (uint64_t)(seed + threadIdx.x + 2) << 40; uint64_t a3 = (uint64_t)(seed + threadIdx.x + 3) << 40; uint64_t a4 = (uint64_t)(seed + threadIdx.x + 4) << 40; uint64_t a5 = (uint64_t)(seed + threadIdx.x + 5) << 40; uint64_t a6 = (uint64_t)(seed + threadIdx.x + 6) << 40; uint64_t a7 = (uint64_t)(seed + threadIdx.x + 7) << 40;
    uint32_t sh = (seed & 1) | 28;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_lshrrev_b64 %0, %8, %0\n v_lshrrev_b64 %1, %8, %1\n v_lshrrev_b64 %2, %8, %2\n v_lshrrev_b64 %3, %8, %3\n v_lshrrev_b64 %4, %8, %4\n v_lshrrev_b64 %5, %8, %5\n v_lshrrev_b64 %6, %8, %6\n v_lshrrev_b64 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_alignbit(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1, sh = 29;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_alignbit_b32 %0, %8, %0, %9\n v_alignbit_b32 %1, %8, %1, %9\n v_alignbit_b32 %2, %8, %2, %9\n v_alignbit_b32 %3, %8, %3, %9\n v_alignbit_b32 %4, %8, %4, %9\n v_alignbit_b32 %5, %8, %5, %9\n v_alignbit_b32 %6, %8, %6, %9\n v_alignbit_b32 %7, %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(sh));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_and_b32(uint32_t* out, uint32_t seed) {
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 0x1fffffff;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_and_b32_e32 %0, %0, %8\n v_and_b32_e32 %1, %1, %8\n v_and_b32_e32 %2, %2, %8\n v_and_b32_e32 %3, %3, %8\n v_and_b32_e32 %4, %4, %8\n v_and_b32_e32 %5, %5, %8\n v_and_b32_e32 %6, %6, %8\n v_and_b32_e32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_shr_inline(uint32_t* out, uint32_t seed) {  // VOP2 e32 with an INLINE CONSTANT as src0
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0) | 0x80000000u; uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1) | 0x80000000u; uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2) | 0x80000000u; uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3) | 0x80000000u; uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4) | 0x80000000u; uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5) | 0x80000000u; uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6) | 0x80000000u; uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7) | 0x80000000u;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_add_u32_e32 %0, 3, %0\n v_add_u32_e32 %1, 3, %1\n v_add_u32_e32 %2, 3, %2\n v_add_u32_e32 %3, 3, %3\n v_add_u32_e32 %4, 3, %4\n v_add_u32_e32 %5, 3, %5\n v_add_u32_e32 %6, 3, %6\n v_add_u32_e32 %7, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_and_literal(uint32_t* out, uint32_t seed) {  // VOP2 e32 with a 32-bit LITERAL as src0
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_add_u32_e32 %0, 0x1fffffff, %0\n v_add_u32_e32 %1, 0x1fffffff, %1\n v_add_u32_e32 %2, 0x1fffffff, %2\n v_add_u32_e32 %3, 0x1fffffff, %3\n v_add_u32_e32 %4, 0x1fffffff, %4\n v_add_u32_e32 %5, 0x1fffffff, %5\n v_add_u32_e32 %6, 0x1fffffff, %6\n v_add_u32_e32 %7, 0x1fffffff, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_cndmask_vcc(uint32_t* out, uint32_t seed) {  // VOP2 e32 v_cndmask (mask = VCC, implicit)
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t y = seed * 7 + 1;
    asm volatile("v_cmp_gt_u32_e32 vcc, 32, %0" : : "v"(threadIdx.x & 63) : "vcc");
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_bpermute(uint32_t* out, uint32_t seed) {  // cross-lane exchange as __shfl_xor compiles it (ds_bpermute_b32)
    uint32_t a0 = (uint32_t)(seed + threadIdx.x + 0); uint32_t a1 = (uint32_t)(seed + threadIdx.x + 1); uint32_t a2 = (uint32_t)(seed + threadIdx.x + 2); uint32_t a3 = (uint32_t)(seed + threadIdx.x + 3); uint32_t a4 = (uint32_t)(seed + threadIdx.x + 4); uint32_t a5 = (uint32_t)(seed + threadIdx.x + 5); uint32_t a6 = (uint32_t)(seed + threadIdx.x + 6); uint32_t a7 = (uint32_t)(seed + threadIdx.x + 7);
    uint32_t addr = ((threadIdx.x ^ 8) & 63) << 2;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_mov_b64(uint32_t* out, uint32_t seed) {
    uint64_t a0 = (uint64_t)(seed + threadIdx.x + 0); uint64_t a1 = (uint64_t)(seed + threadIdx.x + 1); uint64_t a2 = (uint64_t)(seed + threadIdx.x + 2); uint64_t a3 = (uint64_t)(seed + threadIdx.x + 3); uint64_t a4 = (uint64_t)(seed + threadIdx.x + 4); uint64_t a5 = (uint64_t)(seed + threadIdx.x + 5); uint64_t a6 = (uint64_t)(seed + threadIdx.x + 6); uint64_t a7 = (uint64_t)(seed + threadIdx.x + 7);
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_lshl_add64(uint32_t* out, uint32_t seed) {
    uint64_t a0 = (uint64_t)(seed + threadIdx.x + 0); uint64_t a1 = (uint64_t)(seed + threadIdx.x + 1); uint64_t a2 = (uint64_t)(seed + threadIdx.x + 2); uint64_t a3 = (uint64_t)(seed + threadIdx.x + 3); uint64_t a4 = (uint64_t)(seed + threadIdx.x + 4); uint64_t a5 = (uint64_t)(seed + threadIdx.x + 5); uint64_t a6 = (uint64_t)(seed + threadIdx.x + 6); uint64_t a7 = (uint64_t)(seed + threadIdx.x + 7);
    uint64_t y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void k_mad_addc(uint32_t* out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x; uint32_t h = 0; uint32_t x = seed * 3 + threadIdx.x, y = seed * 7 + 1;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n v_mad_u64_u32 %0, s[22:23], %2, %3, %0\n v_mad_u64_u32 %0, s[24:25], %2, %3, %0\n v_mad_u64_u32 %0, s[26:27], %2, %3, %0\n"
                     "v_addc_co_u32_e64 %1, vcc, 0, %1, s[20:21]\n v_addc_co_u32_e64 %1, vcc, 0, %1, s[22:23]\n v_addc_co_u32_e64 %1, vcc, 0, %1, s[24:25]\n v_addc_co_u32_e64 %1, vcc, 0, %1, s[26:27]\n"
                     "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n v_mad_u64_u32 %0, s[22:23], %2, %3, %0\n v_mad_u64_u32 %0, s[24:25], %2, %3, %0\n v_mad_u64_u32 %0, s[26:27], %2, %3, %0\n"
                     "v_addc_co_u32_e64 %1, vcc, 0, %1, s[20:21]\n v_addc_co_u32_e64 %1, vcc, 0, %1, s[22:23]\n v_addc_co_u32_e64 %1, vcc, 0, %1, s[24:25]\n v_addc_co_u32_e64 %1, vcc, 0, %1, s[26:27]\n"
                     : "+v"(a0), "+v"(h) : "v"(x), "v"(y) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 ^ h;
}


#define PROBE8(NAME, TEXT, CLOB)                                                                              \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                                                      \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int it = 0; it < ITERS; it++) {                                                                  \
            asm volatile(TEXT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : CLOB); \
        }                                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                   \
    }
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define I_ADDC64(n) "v_addc_co_u32_e64 %" #n ", vcc, 0, %" #n ", s[20:21]\n"
#define I_ADDC32(n) "v_addc_co_u32_e32 %" #n ", vcc, 0, %" #n ", vcc\n"
#define I_ADDS(n) "v_add_u32_e32 %" #n ", s20, %" #n "\n"
#define I_CNDM(n) "v_cndmask_b32_e64 %" #n ", %" #n ", 1, s[20:21]\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %" #n ", 1\n"
PROBE8(k_addc_e64, R8(I_ADDC64), "vcc")
PROBE8(k_addc_e32, R8(I_ADDC32), "vcc")
PROBE8(k_add_sgpr, R8(I_ADDS), "vcc")
PROBE8(k_cndmask_s, R8(I_CNDM), "vcc")
PROBE8(k_add3, R8(I_ADD3), "vcc")

template <class P, int IMPL>
__global__ void k_femul(const Fe<P>* in, Fe<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<P> x = in[2 * i], y = in[2 * i + 1];
    for (int k = 0; k < iters; k++) {
        if (IMPL == 0) x = fe_mul_cios<P>(x, y);
        else if (IMPL == 1) x = fe_mul_fips<P>(x, y);
        else if (IMPL == 2) x = fe_mul_call<P>(x, y);
        else x = fe_mul_asm<P>(x, y);
    }
    out[i] = x;
}
// radix-2^29 layer: in/out in the C-ABI Montgomery(2^256) form, the chain runs in F29
template <class P>
__global__ void k_f29mul(const Fe<P>* in, Fe<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F29<P> x = f29_from_mont256<P>(in[2 * i]), y = f29_from_mont256<P>(in[2 * i + 1]);
    for (int k = 0; k < iters; k++) x = f29_mul<P>(x, y);
    out[i] = f29_to_mont256<P>(x);
}
template <class P>
__global__ void k_f29dotmul(const Fe<P>* in, Fe<P>* out, int iters) {  // one product as a one-term lazy row (compiler-scheduled mads)
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F29<P> x = f29_from_mont256<P>(in[2 * i]), y = f29_from_mont256<P>(in[2 * i + 1]);
    for (int k = 0; k < iters; k++) { Dot29<P> a; dot29_init<P>(a); dot29_mac<P>(a, x, y); x = dot29_finish<P>(a); }
    out[i] = f29_to_mont256<P>(x);
}
template <class P>
__global__ void k_f29dot2(const Fe<P>* in, Fe<P>* out, int iters) {  // two-term row: x = (x*y + y*y) / R
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F29<P> x = f29_from_mont256<P>(in[2 * i]), y = f29_from_mont256<P>(in[2 * i + 1]);
    for (int k = 0; k < iters; k++) { Dot29<P> a; dot29_init<P>(a); dot29_mac<P>(a, x, y); dot29_mac<P>(a, y, y); x = dot29_finish<P>(a); }
    out[i] = f29_to_mont256<P>(x);
}
template <class P>
__global__ void k_f29sqr(const Fe<P>* in, Fe<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F29<P> x = f29_from_mont256<P>(in[2 * i]);
    for (int k = 0; k < iters; k++) x = f29_sqr<P>(x);
    out[i] = f29_to_mont256<P>(x);
}
template <class P>
__global__ void k_f29addsub(const Fe<P>* in, Fe<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F29<P> x = f29_from_mont256<P>(in[2 * i]), y = f29_from_mont256<P>(in[2 * i + 1]);
    for (int k = 0; k < iters; k++) { x = f29_carry<P>(f29_sub<P>(x, y)); y = f29_carry<P>(f29_add<P>(y, x)); }
    out[i] = f29_to_mont256<P>(f29_add<P>(x, y));
}
template <class P>
__global__ void k_feadd(const Fe<P>* in, Fe<P>* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<P> x = in[2 * i], y = in[2 * i + 1];
    for (int k = 0; k < iters; k++) { x = fe_add<P>(x, y); y = fe_sub<P>(y, x); }
    out[i] = fe_add<P>(x, y);
}

// ---- the FP64-FMA field multiplier experiment (round-3 verdict, item 4) ---------------------------------------------------------
// Pallas Fp in five 52-bit limbs held as doubles; a 52 x 52 -> 104-bit product by the two-FMA split under round-toward-zero
// (Emmart, Zheng, Weems: "Faster modular exponentiation using double precision floating point arithmetic on the GPU", ARITH 2018):
//     hi = fma(a, b, 2^104)            mantissa = floor(ab / 2^52)
//     lo = fma(a, b, (2^104 + 2^52) - hi)   mantissa = ab mod 2^52 (exact)
// the mantissas are accumulated as 64-bit integers (v_lshl_add_u64), the exponent biases are taken off once per column.  Montgomery
// form with R' = 2^260, operand scanning, q_i = t_0 * (-p^-1) mod 2^52 by the same split; p's 52-bit limbs are
// {0xd30ed00000001, 0xfc094cf91b992, 0x224698, 0, 2^46}: 4 of 5 non-zero.  Per product: 25 + 5 + 20 = 50 limb products of
// 2 FMAs + 1 FP add + 2 integer adds each, all quarter-rate instructions like v_mad_u64_u32 (which covers 32 x 32 + 64 in ONE).
__device__ __forceinline__ void fp64_split(double a, double b, unsigned long long& hi_acc, unsigned long long& lo_acc) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    double hi, lo, sub;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(hi) : "v"(a), "v"(b), "v"(C1));
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(sub) : "v"(C2), "v"(hi));
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(lo) : "v"(a), "v"(b), "v"(sub));
    hi_acc += (unsigned long long)__double_as_longlong(hi);
    lo_acc += (unsigned long long)__double_as_longlong(lo);
}
__device__ __forceinline__ double fp64_from52(unsigned long long m) {  // m < 2^52 -> the double m, exactly
    return __longlong_as_double((long long)(m | 0x4330000000000000ull)) - 0x1p52;
}
struct Fp64x5 { double l[5]; };
__device__ __forceinline__ Fp64x5 fp64_mont_mul_pallas(const Fp64x5& a, const Fp64x5& b) {
    const unsigned long long M52 = (1ull << 52) - 1, B_HI = 0x4670000000000000ull /* bits(2^104) */, B_LO = 0x4330000000000000ull /* bits(2^52) */;
    const double P0 = (double)0xd30ed00000001ull, P1 = (double)0xfc094cf91b992ull, P2 = (double)0x224698ull, P4 = 0x1p46, NINV = (double)0xd30ecffffffffull;
    unsigned long long t[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; i++) {
        unsigned long long hi[5] = {0, 0, 0, 0, 0}, lo[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 5; j++) fp64_split(a.l[j], b.l[i], hi[j], lo[j]);
#pragma unroll
        for (int j = 0; j < 5; j++) { t[j] += lo[j] - B_LO; t[j + 1] += hi[j] - B_HI; }
        unsigned long long qh = 0, ql = 0;
        fp64_split(fp64_from52(t[0] & M52), NINV, qh, ql);
        const double q = fp64_from52((ql - B_LO) & M52);
        unsigned long long h0 = 0, l0 = 0, h1 = 0, l1 = 0, h2 = 0, l2 = 0, h4 = 0, l4 = 0;
        fp64_split(q, P0, h0, l0);
        fp64_split(q, P1, h1, l1);
        fp64_split(q, P2, h2, l2);
        fp64_split(q, P4, h4, l4);
        t[0] += l0 - B_LO; t[1] += (h0 - B_HI) + (l1 - B_LO); t[2] += (h1 - B_HI) + (l2 - B_LO); t[3] += h2 - B_HI; t[4] += l4 - B_LO; t[5] += h4 - B_HI;
        // t[0] is now a multiple of 2^52: shift the accumulator down one limb
        t[1] += t[0] >> 52;
        t[0] = t[1]; t[1] = t[2]; t[2] = t[3]; t[3] = t[4]; t[4] = t[5]; t[5] = 0;
    }
    // carry the columns into 52-bit limbs; the value is < 2 p
    unsigned long long c = 0, r[5];
#pragma unroll
    for (int j = 0; j < 5; j++) { const unsigned long long v = t[j] + c; r[j] = v & M52; c = v >> 52; }
    // one conditional subtraction of p (limbs as integers)
    const unsigned long long PL[5] = {0xd30ed00000001ull, 0xfc094cf91b992ull, 0x224698ull, 0ull, 0x400000000000ull};
    unsigned long long d[5];
    long long br = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { const long long v = (long long)r[j] - (long long)PL[j] + br; d[j] = (unsigned long long)v & M52; br = v >> 52; }
    Fp64x5 o;
#pragma unroll
    for (int j = 0; j < 5; j++) o.l[j] = fp64_from52(br < 0 ? r[j] : d[j]);
    return o;
}
__device__ __forceinline__ Fp64x5 fp64_from_fe(const Fe<PallasFp>& x) {
    unsigned long long w[4];
    for (int i = 0; i < 4; i++) w[i] = (unsigned long long)x.l[2 * i] | ((unsigned long long)x.l[2 * i + 1] << 32);
    const unsigned long long M52 = (1ull << 52) - 1;
    Fp64x5 o;
    o.l[0] = fp64_from52(w[0] & M52);
    o.l[1] = fp64_from52(((w[0] >> 52) | (w[1] << 12)) & M52);
    o.l[2] = fp64_from52(((w[1] >> 40) | (w[2] << 24)) & M52);
    o.l[3] = fp64_from52(((w[2] >> 28) | (w[3] << 36)) & M52);
    o.l[4] = fp64_from52(w[3] >> 16);
    return o;
}
__device__ __forceinline__ Fe<PallasFp> fp64_to_fe(const Fp64x5& x) {
    unsigned long long m[5];
    for (int j = 0; j < 5; j++) m[j] = (unsigned long long)__double_as_longlong(x.l[j] + 0x1p52) & ((1ull << 52) - 1);
    unsigned long long w[4] = {m[0] | (m[1] << 52), (m[1] >> 12) | (m[2] << 40), (m[2] >> 24) | (m[3] << 28), (m[3] >> 36) | (m[4] << 16)};
    Fe<PallasFp> o;
    for (int i = 0; i < 4; i++) { o.l[2 * i] = (uint32_t)w[i]; o.l[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return o;
}
// the same chain as k_femul: x <- x * y / 2^256; with R' = 2^260 the multiplier enters as 16 y, so that x * (16 y) / 2^260 = x * y / 2^256
__global__ void k_fp64mul(const Fe<PallasFp>* in, Fe<PallasFp>* out, int iters) {
    // MODE.FP_ROUND[3:2] (f64 / f16) = round toward zero.  As asm volatile, like the FMAs: the compiler keeps volatile asm statements in
    // order among themselves, but moved the s_setreg BUILTIN (both of them) above the loop of asm FMAs, which then ran in round-to-nearest
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<PallasFp> y16 = in[2 * i + 1];
    for (int k = 0; k < 4; k++) y16 = fe_add<PallasFp>(y16, y16);
    Fp64x5 x = fp64_from_fe(in[2 * i]);
    const Fp64x5 y = fp64_from_fe(y16);
    for (int k = 0; k < iters; k++) x = fp64_mont_mul_pallas(x, y);
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0");
    out[i] = fp64_to_fe(x);
}

static double time_kernel(std::function<void()> f, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 16, threads = 256;
    uint32_t* d_out; CK(hipMalloc(&d_out, (size_t)blocks * threads * 4));
    double waves = (double)blocks * threads / 64;
#define RUN_PROBE(K, PER_ITER, LABEL) { double ms = time_kernel([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u); }); \
        double winstr = waves * ITERS * PER_ITER; printf("%-22s %8.3f ms  %8.2f G wave-instr/s  %7.2f T lane-ops/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", LABEL, ms, winstr / ms / 1e6, winstr * 64 / ms / 1e9, \
        2.4e9 * (ms / 1e3) * prop.multiProcessorCount * 4 / winstr); }
    RUN_PROBE(k_mad64, 8, "v_mad_u64_u32");
    RUN_PROBE(k_mul_lo, 8, "v_mul_lo_u32");
    RUN_PROBE(k_mul_hi, 8, "v_mul_hi_u32");
    RUN_PROBE(k_mul_u24, 8, "v_mul_u32_u24");
    RUN_PROBE(k_add_u32, 8, "v_add_u32");
    RUN_PROBE(k_xor, 8, "v_xor_b32");
    RUN_PROBE(k_fma64, 8, "v_fma_f64");
    RUN_PROBE(k_lshl_add64, 8, "v_lshl_add_u64");
    RUN_PROBE(k_lshr64, 8, "v_lshrrev_b64");
    RUN_PROBE(k_alignbit, 8, "v_alignbit_b32");
    RUN_PROBE(k_and_b32, 8, "v_and_b32");
    RUN_PROBE(k_mov_b64, 8, "v_mov_b64");
    RUN_PROBE(k_shr_inline, 8, "v_add_u32 e32 inline const");
    RUN_PROBE(k_and_literal, 8, "v_add_u32 e32 literal");
    RUN_PROBE(k_cndmask_vcc, 8, "v_cndmask e32 (vcc)");
    RUN_PROBE(k_bpermute, 8, "ds_bpermute_b32");
    RUN_PROBE(k_mad_addc, 16, "mad+addc (dep chain)");
    RUN_PROBE(k_addc_e64, 8, "v_addc_co e64 sgpr-carry");
    RUN_PROBE(k_addc_e32, 8, "v_addc_co e32 vcc (dep)");
    RUN_PROBE(k_add_sgpr, 8, "v_add_u32 sgpr operand");
    RUN_PROBE(k_cndmask_s, 8, "v_cndmask e64 sgpr mask");
    RUN_PROBE(k_add3, 8, "v_add3_u32 (VOP3)");

    // field multiplier variants
    size_t n = (size_t)blocks * threads;
    std::vector<uint32_t> h(n * 16);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s; }
    for (size_t i = 0; i < n * 2; i++) h[i * 8 + 7] &= 0x1fffffff;  // < every modulus
    void *d_in, *d_o0, *d_o1, *d_o2;
    CK(hipMalloc(&d_in, n * 64)); CK(hipMalloc(&d_o0, n * 32)); CK(hipMalloc(&d_o1, n * 32)); CK(hipMalloc(&d_o2, n * 32));
    CK(hipMemcpy(d_in, h.data(), n * 64, hipMemcpyHostToDevice));
    const int MI = 512;
#define RUN_MUL(P, IMPL, OUT, LABEL) { double ms = time_kernel([&] { hipLaunchKernelGGL((k_femul<P, IMPL>), dim3(blocks), dim3(threads), 0, 0, (const Fe<P>*)d_in, (Fe<P>*)OUT, MI); }); \
        printf("%-28s %8.3f ms  %8.2f G field-mul/s\n", LABEL, ms, (double)n * MI / ms / 1e6); }
    RUN_MUL(PallasFp, 0, d_o0, "fe_mul Pallas cios(compiler)");
    RUN_MUL(PallasFp, 1, d_o1, "fe_mul Pallas fips(asm)");
    RUN_MUL(PallasFp, 2, d_o2, "fe_mul Pallas default noinline");
    RUN_MUL(PallasFp, 3, d_o1, "fe_mul Pallas asm-block");
    {
        std::vector<uint32_t> r0(n * 8), r1(n * 8), r2(n * 8);
        CK(hipMemcpy(r0.data(), d_o0, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d_o1, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), d_o2, n * 32, hipMemcpyDeviceToHost));
        size_t bad1 = 0, bad2 = 0;
        for (size_t i = 0; i < n * 8; i++) { bad1 += r0[i] != r1[i]; bad2 += r0[i] != r2[i]; }
        printf("  Pallas: fips vs cios mismatching words: %zu, noinline vs cios: %zu (of %zu)\n", bad1, bad2, n * 8);
    }
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_f29mul<PallasFp>), dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o1, MI); });
      printf("%-28s %8.3f ms  %8.2f G field-mul/s\n", "f29_mul Pallas (radix 2^29)", ms, (double)n * MI / ms / 1e6);
      std::vector<uint32_t> r0(n * 8), r1(n * 8);
      CK(hipMemcpy(r0.data(), d_o0, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d_o1, n * 32, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < n * 8; i++) bad += r0[i] != r1[i];
      printf("  Pallas: radix-2^29 chain vs cios chain mismatching words: %zu (of %zu)\n", bad, n * 8); }
    { double ms = time_kernel([&] { hipLaunchKernelGGL(k_fp64mul, dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o2, MI); });
      printf("%-28s %8.3f ms  %8.2f G field-mul/s\n", "fp64-FMA mul Pallas (5x52)", ms, (double)n * MI / ms / 1e6);
      std::vector<uint32_t> r0(n * 8), r1(n * 8);
      CK(hipMemcpy(r0.data(), d_o0, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d_o2, n * 32, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < n * 8; i++) bad += r0[i] != r1[i];
      printf("  Pallas: fp64-FMA chain vs cios chain mismatching words: %zu (of %zu)\n", bad, n * 8); }
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_f29dotmul<PallasFp>), dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o2, MI); });
      printf("%-28s %8.3f ms  %8.2f G field-mul/s\n", "f29 1-term lazy row (C++)", ms, (double)n * MI / ms / 1e6);
      std::vector<uint32_t> r0(n * 8), r1(n * 8);
      CK(hipMemcpy(r0.data(), d_o0, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d_o2, n * 32, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < n * 8; i++) bad += r0[i] != r1[i];
      printf("  Pallas: 1-term row chain vs cios chain mismatching words: %zu\n", bad); }
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_f29dot2<PallasFp>), dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o2, MI); });
      printf("%-28s %8.3f ms  %8.2f G rows/s (2 products each)\n", "f29 2-term lazy row (C++)", ms, (double)n * MI / ms / 1e6); }
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_f29sqr<PallasFp>), dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o2, MI); });
      printf("%-28s %8.3f ms  %8.2f G field-sqr/s\n", "f29_sqr Pallas (radix 2^29)", ms, (double)n * MI / ms / 1e6); }
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_f29addsub<PallasFp>), dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o2, MI); });
      printf("%-28s %8.3f ms  %8.2f G (add|sub)+carry /s\n", "f29 sub+carry, add+carry", ms, (double)n * MI * 2 / ms / 1e6); }
    RUN_MUL(Bn254Fr, 0, d_o0, "fe_mul BN254 cios(compiler)");
    RUN_MUL(Bn254Fr, 1, d_o1, "fe_mul BN254 fips(asm)");
    RUN_MUL(Bn254Fr, 3, d_o1, "fe_mul BN254 asm-block");
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_f29mul<Bn254Fr>), dim3(blocks), dim3(threads), 0, 0, (const Fe<Bn254Fr>*)d_in, (Fe<Bn254Fr>*)d_o2, MI); });
      printf("%-28s %8.3f ms  %8.2f G field-mul/s\n", "f29_mul BN254 (radix 2^29)", ms, (double)n * MI / ms / 1e6);
      std::vector<uint32_t> r0(n * 8), r1(n * 8);
      CK(hipMemcpy(r0.data(), d_o0, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d_o2, n * 32, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < n * 8; i++) bad += r0[i] != r1[i];
      printf("  BN254: radix-2^29 chain vs cios chain mismatching words: %zu (of %zu)\n", bad, n * 8); }
    {
        std::vector<uint32_t> r0(n * 8), r1(n * 8);
        CK(hipMemcpy(r0.data(), d_o0, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d_o1, n * 32, hipMemcpyDeviceToHost));
        size_t bad1 = 0;
        for (size_t i = 0; i < n * 8; i++) bad1 += r0[i] != r1[i];
        printf("  BN254: fips vs cios mismatching words: %zu (of %zu)\n", bad1, n * 8);
    }
    { double ms = time_kernel([&] { hipLaunchKernelGGL((k_feadd<PallasFp>), dim3(blocks), dim3(threads), 0, 0, (const Fe<PallasFp>*)d_in, (Fe<PallasFp>*)d_o0, MI); });
      printf("%-28s %8.3f ms  %8.2f G field-add-or-sub/s\n", "fe_add+fe_sub Pallas", ms, (double)n * MI * 2 / ms / 1e6); }
    return 0;
}

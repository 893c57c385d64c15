// mds_mfma.hip - ONE bounded experiment: Poseidon's full-round MDS layer (t = 9: s <- M s, 81 products by CONSTANTS
// per state) on the matrix cores instead of the integer VALU.  Tooling, not product.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 bench_tools/mds_mfma.hip -o bench_tools/mds_mfma
//
// A product by a constant c is a Toeplitz matrix-vector product over 8-bit limbs: z_k = sum_{a+b=k} s[a] c[b].  For the whole layer
//     Z (10 x 64 rows: output element j, column k)  =  A (640 x 320, int8 constants)  x  S (320 x N states, int8 state bytes)
// with v_mfma_i32_32x32x32_i8 (32 x 32 x 32, i32 accumulation; |z_k| < 2^25).  The layout needs no cross-lane data movement:
//   * a state is held by TWO lanes (l, l + 32): half 0 owns elements 0..4, half 1 owns 5..8; B's lane-half h supplies 16 bytes of an
//     element IT owns (the K order is ours to choose: A and B use the same (half, byte) -> k map whatever the hardware's numbering);
//   * the rows of A are ordered so that the 16 accumulator registers a lane gets from each of 4 M-tiles are the 64 columns of ONE output
//     element the lane's half owns: C/D layout row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)  (cdna_hip_programming.md, 32x32 shapes);
//   * signed operands: state bytes are offset (s' = s - 128, one XOR per word), constants use balanced digits in [-128, 127]; the
//     offset's correction, a per-column bias that keeps every column sum non-negative and the multiple of p that cancels the bias all
//     fold into the accumulators' INITIAL values (constants per (j, k)): nothing is paid for them at run time;
//   * a lane then turns its 64 column sums into the 17 radix-2^29 columns of field29.cuh's lazy row and runs the SAME single Montgomery
//     reduction (dot29_finish): 64 shift-adds + 54 mads per output element instead of 81 x 9 + 54 mads.
// A fragments (16 descending digits of one constant at a lane-dependent offset) come from LDS: every constant reversed, zero padded and
// stored at the four byte alignments (81 x 4 x 96 B = 31 KB), so each lane reads four aligned dwords per MFMA and reuses them for two
// state tiles (64 states per wave; one tile per fragment load would saturate the LDS at four waves per CU).
// Checked bit-exact (canonical outputs) against poseidon29.cuh's lazy rows on the same constants, conversions both ways included.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#define LURK_MUL_FORCE_INLINE_OFF
#include "../lurk_beta_amd/csrc/field.cuh"
#include "../lurk_beta_amd/csrc/field29.cuh"
using namespace lurk;
using P = PallasFq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int T = 9;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- the VALU form (what poseidon29_dense does): one state per lane, nine lazy rows -------------------------------------------------
__global__ __launch_bounds__(256) void k_mds_valu(const uint32_t* __restrict__ mat /*81 x 9 limbs*/, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                  size_t n, int iters) {
    __shared__ uint32_t m[81 * 9];
    for (int i = threadIdx.x; i < 81 * 9; i += 256) m[i] = mat[i];
    __syncthreads();
    const size_t id = blockIdx.x * (size_t)256 + threadIdx.x;
    if (id >= n) return;
    F29<P> s[T];
    for (int e = 0; e < T; e++)
        for (int l = 0; l < 9; l++) s[e].l[l] = in[(id * T + e) * 9 + l];
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        F29<P> u[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            Dot29<P> A;
            dot29_init<P>(A);
#pragma unroll
            for (int i = 0; i < T; i++) {
                if (i == 4) dot29_carry<P>(A);
                const uint32_t* c = m + (j * T + i) * 9;
                asm volatile("" : "+v"(c));
                F29<P> cc;
#pragma unroll
                for (int l = 0; l < 9; l++) cc.l[l] = c[l];
                dot29_mac<P>(A, s[i], cc);
            }
            u[j] = dot29_finish<P>(A);
        }
#pragma unroll
        for (int j = 0; j < T; j++) s[j] = u[j];
    }
    for (int e = 0; e < T; e++)
        for (int l = 0; l < 9; l++) out[(id * T + e) * 9 + l] = s[e].l[l];
}

// ---- the MFMA form ------------------------------------------------------------------------------------------------------------------
constexpr int NT = 2;             // state tiles per wave (32 states each)
constexpr int REV_BYTES = 96;     // one reversed, padded constant: R[32 + t] = digit[31 - t]
constexpr int LDS_CONST = 82 * 4 * REV_BYTES;  // 81 constants + one all-zero record, four byte alignments each
constexpr int LDS_CINIT = 10 * 64 * 4;         // initial accumulators per (output element, column); element 9 = padding (zero)

struct MdsTables {
    const uint8_t* rev;     // [82][4][96]
    const int32_t* cinit;   // [10][64]
};

__device__ __forceinline__ void pack_bytes(const F29<P>& a, uint32_t* w) {  // tight, < 2^256 -> 8 words of offset bytes (s - 128)
    f29_pack<P>(a, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] ^= 0x80808080u;
}

__global__ __launch_bounds__(256) void k_mds_mfma(MdsTables tb, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n_states, int iters) {
    extern __shared__ uint8_t lds[];
    uint8_t* rev = lds;
    int32_t* cinit = reinterpret_cast<int32_t*>(lds + LDS_CONST);
    uint32_t* nwl = reinterpret_cast<uint32_t*>(lds + LDS_CONST + LDS_CINIT);  // [NT][5][9][256 threads]
    for (int i = threadIdx.x; i < LDS_CONST / 4; i += 256) reinterpret_cast<uint32_t*>(rev)[i] = reinterpret_cast<const uint32_t*>(tb.rev)[i];
    for (int i = threadIdx.x; i < 10 * 64; i += 256) cinit[i] = tb.cinit[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const size_t wave_id = blockIdx.x * (size_t)4 + wave;
    const int own = half == 0 ? 5 : 4;   // elements this lane owns: half 0 -> 0..4, half 1 -> 5..8
    // A-side constants of this lane: its row of every M-tile
    const int rho = lane & 31, h_out = (rho >> 2) & 1, idx = (rho >> 3) * 4 + (rho & 3);
    F29<P> st[NT][5];
    for (int t = 0; t < NT; t++) {
        const size_t sid = (wave_id * NT + t) * 32 + col;
        for (int e = 0; e < 5; e++)
            for (int l = 0; l < 9; l++) st[t][e].l[l] = (sid < n_states && e < own) ? in[(sid * T + half * 5 + e) * 9 + l] : 0u;
    }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        uint32_t by[NT][5][8];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int e = 0; e < 5; e++) pack_bytes(st[t][e], by[t][e]);
        // the new state waits in LDS (9 words per element and lane): 90 registers less across the pair loop
#pragma unroll 1
        for (int p = 0; p < 5; p++) {
            // output elements of this pair: half 0 -> p, half 1 -> 5 + p (p = 4: padding)
            const int Jrow = h_out == 0 ? p : (p < 4 ? 5 + p : 9);   // the output element this lane's A ROW belongs to
            const int Jown = half == 0 ? p : (p < 4 ? 5 + p : 9);    // the output element this lane's ACCUMULATORS hold
            v16i acc[NT][4];
#pragma unroll
            for (int tau = 0; tau < 4; tau++) {
                const v4i* ci = reinterpret_cast<const v4i*>(cinit + Jown * 64 + tau * 16);
                v16i c0;
#pragma unroll
                for (int q = 0; q < 4; q++) { v4i x = ci[q]; c0[4 * q] = x[0]; c0[4 * q + 1] = x[1]; c0[4 * q + 2] = x[2]; c0[4 * q + 3] = x[3]; }
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][tau] = c0;
            }
#pragma unroll
            for (int e = 0; e < 5; e++) {
                // the input element lane-half `half` supplies at this K-step: half 0 -> e, half 1 -> 5 + e (e = 4: padding, A = 0)
                const int I = half == 0 ? e : (e < 4 ? 5 + e : 9);
                const int rec = (Jrow == 9 || I == 9) ? 81 : Jrow * 9 + I;
#pragma unroll
                for (int ch = 0; ch < 2; ch++) {
#pragma unroll
                    for (int tau = 0; tau < 4; tau++) {
                        const int k = 16 * tau + idx;
                        const int x0 = 63 - k + 16 * ch;          // first of the 16 descending digits in the reversed record
                        const uint32_t* src = reinterpret_cast<const uint32_t*>(rev + (rec * 4 + (x0 & 3)) * REV_BYTES + (x0 & ~3));
                        v4i a;
                        a[0] = (int)src[0]; a[1] = (int)src[1]; a[2] = (int)src[2]; a[3] = (int)src[3];
#pragma unroll
                        for (int t = 0; t < NT; t++) {
                            v4i b;
                            b[0] = (int)by[t][e][4 * ch]; b[1] = (int)by[t][e][4 * ch + 1]; b[2] = (int)by[t][e][4 * ch + 2]; b[3] = (int)by[t][e][4 * ch + 3];
                            acc[t][tau] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t][tau], 0, 0, 0);
                        }
                    }
                }
            }
            // 64 non-negative column sums (bits 8k) -> the 17 columns of the lazy row (bits 29m) -> one Montgomery reduction
#pragma unroll
            for (int t = 0; t < NT; t++) {
                Dot29<P> A;
                dot29_init<P>(A);
#pragma unroll
                for (int tau = 0; tau < 4; tau++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int k = 16 * tau + r, bit = 8 * k;
                        const int m = bit / 29 > 16 ? 16 : bit / 29, sh = bit - 29 * m;
                        // one v_mad_u64_u32 per column (the shift as a multiplication by a constant power of two); column 62 lands 32 bits up
                        if (sh < 32) A.c[m] += (uint64_t)(uint32_t)acc[t][tau][r] * (uint64_t)(1u << sh);
                        else A.c[m] += (uint64_t)(uint32_t)acc[t][tau][r] << sh;
                    }
                const F29<P> o = dot29_finish<P>(A);
#pragma unroll
                for (int l = 0; l < 9; l++) nwl[((t * 5 + p) * 9 + l) * 256 + threadIdx.x] = o.l[l];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int e = 0; e < 5; e++)
#pragma unroll
                for (int l = 0; l < 9; l++) st[t][e].l[l] = nwl[((t * 5 + e) * 9 + l) * 256 + threadIdx.x];
    }
    for (int t = 0; t < NT; t++) {
        const size_t sid = (wave_id * NT + t) * 32 + col;
        if (sid >= n_states) continue;
        for (int e = 0; e < own; e++)
            for (int l = 0; l < 9; l++) out[(sid * T + half * 5 + e) * 9 + l] = st[t][e].l[l];
    }
}

// ---- host: constants, tables, comparison -----------------------------------------------------------------------------------------------
typedef unsigned __int128 u128;
struct Big { uint32_t w[20]; };  // up to 640 bits
static Big big_zero() { Big b; memset(&b, 0, sizeof b); return b; }
static void big_add_shifted(Big& b, uint64_t v, int bit) {  // b += v << bit
    int wi = bit / 32, sh = bit % 32;
    u128 x = (u128)v << sh;
    uint64_t carry = 0;
    for (int i = wi; i < 20 && (x || carry); i++) {
        uint64_t s = (uint64_t)b.w[i] + (uint32_t)x + carry;
        b.w[i] = (uint32_t)s;
        carry = s >> 32;
        x >>= 32;
    }
}
static void big_mod_p(const Big& b, uint32_t* out8) {  // b mod p by repeated conditional subtraction from the top (schoolbook, host only)
    // reduce with 8 x 32 Montgomery-free arithmetic: process the words from the top, r = (r * 2^32 + w) mod p
    Fe<P> r = fe_zero<P>();
    for (int i = 19; i >= 0; i--) {
        for (int d = 0; d < 32; d++) r = fe_add<P>(r, r);  // r * 2^32 mod p (canonical doubling)
        Fe<P> w = fe_zero<P>();
        w.l[0] = b.w[i];
        r = fe_add<P>(r, w);
    }
    for (int i = 0; i < 8; i++) out8[i] = r.l[i];
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return (uint32_t)seed; };
    // 81 constants: canonical values < p (the image's c * 2^261 mod p are of this kind)
    std::vector<Fe<P>> C(81);
    for (auto& c : C) {
        for (int i = 0; i < 8; i++) c.l[i] = rnd();
        c.l[7] &= 0x3fffffffu;  // < 2^254 < p
    }
    std::vector<uint32_t> mat29(81 * 9);
    for (int c = 0; c < 81; c++) { F29<P> f = f29_from_plain<P>(C[c].l); for (int l = 0; l < 9; l++) mat29[c * 9 + l] = f.l[l]; }
    // balanced digits
    std::vector<int> dig(81 * 32);
    for (int c = 0; c < 81; c++) {
        int carry = 0;
        for (int b = 0; b < 32; b++) {
            int v = (int)((C[c].l[b / 4] >> (8 * (b % 4))) & 0xff) + carry;
            if (v >= 128) { v -= 256; carry = 1; } else carry = 0;
            dig[c * 32 + b] = v;
        }
        if (carry) { printf("constant too large for 32 balanced digits\n"); return 1; }
    }
    std::vector<uint8_t> rev((size_t)82 * 4 * REV_BYTES, 0);
    for (int c = 0; c < 81; c++)
        for (int sft = 0; sft < 4; sft++)
            for (int y = 0; y < REV_BYTES; y++) {
                const int x = y + sft;  // Rs[sft][y] = R[y + sft], R[32 + t] = digit[31 - t]
                int v = 0;
                if (x >= 32 && x < 64) v = dig[c * 32 + (31 - (x - 32))];
                rev[((size_t)c * 4 + sft) * REV_BYTES + y] = (uint8_t)(int8_t)v;
            }
    std::vector<int32_t> cinit(10 * 64, 0);
    for (int j = 0; j < 9; j++) {
        long long corr[64], beta[64];
        Big D = big_zero();
        for (int k = 0; k < 64; k++) {
            long long sum = 0, neg = 0;
            for (int i = 0; i < 9; i++)
                for (int b = (k > 31 ? k - 31 : 0); b <= (k < 31 ? k : 31); b++) {
                    const int d = dig[(j * 9 + i) * 32 + b];
                    sum += d;
                    if (d < 0) neg += -d;
                }
            corr[k] = 128 * sum;    // the s' = s - 128 offset
            beta[k] = 255 * neg;    // keeps the column non-negative
            big_add_shifted(D, (uint64_t)beta[k], 8 * k);
        }
        uint32_t dm[8];
        big_mod_p(D, dm);
        Fe<P> d; for (int i = 0; i < 8; i++) d.l[i] = dm[i];
        Fe<P> e = fe_neg<P>(d);  // (-D) mod p, canonical
        for (int k = 0; k < 64; k++) {
            long long v = corr[k] + beta[k] + (k < 32 ? (long long)((e.l[k / 4] >> (8 * (k % 4))) & 0xff) : 0);
            if (v < -(1ll << 30) || v > (1ll << 30)) { printf("cinit out of range\n"); return 1; }
            cinit[j * 64 + k] = (int32_t)v;
        }
    }
    // states
    const size_t n = (size_t)prop.multiProcessorCount * 4 * NT * 32 * 8;  // 8 waves' worth per SIMD
    std::vector<uint32_t> in(n * T * 9);
    for (size_t s = 0; s < n * T; s++) {
        uint32_t w[8];
        for (int i = 0; i < 8; i++) w[i] = rnd();
        w[7] &= 0x7fffffffu;  // < 2^255: what an S-box output is bounded by (DESIGN: product outputs < 2^251 + p)
        F29<P> f = f29_from_plain<P>(w);
        for (int l = 0; l < 9; l++) in[s * 9 + l] = f.l[l];
    }
    uint32_t *d_in, *d_o1, *d_o2, *d_mat; uint8_t* d_rev; int32_t* d_ci;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_o1, in.size() * 4)); CK(hipMalloc(&d_o2, in.size() * 4));
    CK(hipMalloc(&d_mat, mat29.size() * 4)); CK(hipMalloc(&d_rev, rev.size())); CK(hipMalloc(&d_ci, cinit.size() * 4));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mat, mat29.data(), mat29.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rev, rev.data(), rev.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ci, cinit.data(), cinit.size() * 4, hipMemcpyHostToDevice));
    MdsTables tb{d_rev, d_ci};
    const size_t lds = LDS_CONST + LDS_CINIT + (size_t)NT * 5 * 9 * 256 * 4;
    CK(hipFuncSetAttribute((const void*)k_mds_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned blocks_v = (unsigned)((n + 255) / 256), blocks_m = (unsigned)((n + 4 * NT * 32 - 1) / (4 * NT * 32));
    auto time_kernel = [&](std::function<void()> f) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        f(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int r = 0; r < 3; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
        return best;
    };
    auto canon_equal = [&](int iters) {
        hipLaunchKernelGGL(k_mds_valu, dim3(blocks_v), dim3(256), 0, 0, d_mat, d_in, d_o1, n, iters);
        hipLaunchKernelGGL(k_mds_mfma, dim3(blocks_m), dim3(256), lds, 0, tb, d_in, d_o2, n, iters);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> a(in.size()), b(in.size());
        CK(hipMemcpy(a.data(), d_o1, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_o2, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t s = 0; s < n * T; s++) {  // canonical value of x / 2^261-domain representative: compare x mod p
            auto canon = [&](const uint32_t* l9, uint32_t* w8) {
                Big v = big_zero();
                for (int l = 0; l < 9; l++) big_add_shifted(v, l9[l], 29 * l);
                big_mod_p(v, w8);
            };
            uint32_t x[8], y[8];
            canon(&a[s * 9], x); canon(&b[s * 9], y);
            bad += memcmp(x, y, 32) != 0;
        }
        return bad;
    };
    printf("bit-exact check (canonical outputs of every element, %zu states): 1 application: %zu mismatches, 3 applications: %zu mismatches\n", n,
           canon_equal(1), canon_equal(3));
    const int IT = 64;
    const double ms_v = time_kernel([&] { hipLaunchKernelGGL(k_mds_valu, dim3(blocks_v), dim3(256), 0, 0, d_mat, d_in, d_o1, n, IT); });
    const double ms_m = time_kernel([&] { hipLaunchKernelGGL(k_mds_mfma, dim3(blocks_m), dim3(256), lds, 0, tb, d_in, d_o2, n, IT); });
    printf("MDS layer t = 9, %zu states x %d applications\n", n, IT);
    printf("  VALU lazy rows (poseidon29_dense form)     %8.3f ms  %8.2f M layers/s\n", ms_v, (double)n * IT / ms_v / 1e3);
    printf("  MFMA i8 32x32x32 + one reduction per row    %8.3f ms  %8.2f M layers/s   speed-up %.2fx\n", ms_m, (double)n * IT / ms_m / 1e3, ms_v / ms_m);
    printf("  MFMA form: %d MFMAs per 64 states and layer = %.1f G MFMA/s = %.0f TOPS int8 (dense 32x32x32 = 65536 ops)\n", 400,
           (double)n / 64 * 400 * IT / ms_m / 1e6, (double)n / 64 * 400 * IT / ms_m / 1e6 * 65536 / 1e3);
    return 0;
}

// ntt.hip - radix-2 NTT over the Pasta fields for gfx950.
//
// No reference counterpart: lurk-beta / arecibo are sum-check based and contain no NTT
// (SURVEY.md section 0.5); parity is against the textbook oracle only ("parity unpinned").
// omega_n = (5^((p-1)/2^32))^(2^(32-log_n)); natural order in, natural order out; the inverse
// transform uses omega^-1 and scales by 1/n.
//
// Two implementations of the same twisted decimation-in-time:
//   log_n >= 12  wave-resident passes (ntt_wave_pass_kernel below): <= 8 stages per pass, a wave owns 256 elements of a 2048-element
//                tile and runs radix-4 rounds entirely in registers; between register rounds its 256 elements are transposed through
//                a swizzled, bank-conflict-free LDS buffer (no lane shuffles, no selects); radix-2^29 arithmetic with a 4p bias, bit
//                reversal folded into the first pass's tile addressing, LDS also holds the twiddle table: 3 passes over memory at 2^24
//   smaller      the LDS-stage kernels (sizes below 2^12 only):
// Shape of the LDS-stage path: decimation-in-time after a bit-reversal permutation, in passes that each fuse several
// radix-2 stages out of an LDS tile:
//   * the permutation is a 32x32 LDS transpose (both the reads and the writes are 1 KiB rows);
//   * pass 1 (stages 1..11) works on 2048 contiguous elements;
//   * a later pass covering stages s0+1..s0+ns takes, per workgroup, C = 2048 / 2^ns neighbouring
//     sub-transforms so that every global row is C x 32 B contiguous, "twists" each element once by
//     omega_N^(t * rev(l)) (N = 2^(s0+ns), t = position inside the already transformed 2^s0 block),
//     after which every stage only needs the 2^(ns-1) local twiddles, kept in LDS;
//   * the last pass writes straight into the caller's buffer (scaled / converted as needed).
#include <memory>
#include <tuple>

#include "common.hpp"
#include "curve29.cuh"
#include "poseidon29.cuh"

namespace lurk {


template <class F>
__device__ Fe<F> ntt_pow(Fe<F> base, uint64_t e) {
    Fe<F> acc = fe_one<F>();
    while (e) {
        if (e & 1) acc = fe_mul<F>(acc, base);
        base = fe_sqr<F>(base);
        e >>= 1;
    }
    return acc;
}

// tw[i] = omega^i, i < n/2 (Montgomery)
template <class F>
__global__ __launch_bounds__(256) void ntt_twiddle_kernel(Fe<F> omega, size_t half, Fe<F>* tw) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < half) tw[i] = ntt_pow<F>(omega, i);
}

// dst[bitrev(i)] = to_mont(src[i])  (out of place), small sizes
template <class F>
__global__ __launch_bounds__(256) void ntt_bitrev_kernel(const Fe<F>* __restrict__ src, Fe<F>* __restrict__ dst, unsigned log_n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >> log_n) return;
    size_t r = log_n ? (__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
    dst[r] = fe_to_mont<F>(src[i]);
}
// same for log_n >= 10 as a tiled transpose: index bits [hi:5][mid][lo:5] -> [rev lo][rev mid][rev hi];
// block = one value of mid; reads rows of 32 consecutive lo, writes rows of 32 consecutive rev(hi)
template <class F>
__global__ __launch_bounds__(256) void ntt_bitrev_tiled_kernel(const Fe<F>* __restrict__ src, Fe<F>* __restrict__ dst, unsigned log_n) {
    __shared__ uint4 lds_raw[32 * 33 * 2];
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(lds_raw);
    const unsigned mbits = log_n - 10;
    const size_t mid = blockIdx.x;
    const size_t rmid = mbits ? (__brevll((unsigned long long)mid) >> (64 - mbits)) : 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned e = r * 256 + threadIdx.x, hi = e >> 5, lo = e & 31;
        sh[lo * 33 + hi] = fe_to_mont<F>(src[((size_t)hi << (log_n - 5)) | (mid << 5) | lo]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned e = r * 256 + threadIdx.x, a = e >> 5, b = e & 31;  // a = rev5(lo), b = rev5(hi)
        unsigned lo = __brev(a) >> 27, hi = __brev(b) >> 27;
        dst[((size_t)a << (log_n - 5)) | (rmid << 5) | b] = sh[lo * 33 + hi];
    }
}

constexpr int NTT_TILE_LOG = 11;      // 2048 elements (64 KiB) per workgroup
constexpr int NTT_PASS_BLOCK = 512;

// One pass: stages s0+1 .. s0+ns.  cbits = log2(C), ns + cbits <= NTT_TILE_LOG.  Tile layout in LDS:
// [l][c] (l = index inside the sub-transform, c = which of the C neighbouring sub-transforms).
template <class F>
__global__ __launch_bounds__(NTT_PASS_BLOCK) void ntt_pass_kernel(const Fe<F>* __restrict__ in, Fe<F>* __restrict__ out,
                                                                    const Fe<F>* __restrict__ tw, unsigned log_n, unsigned s0, unsigned ns,
                                                                    unsigned cbits, Fe<F> scale, int do_scale, int out_canonical) {
    extern __shared__ uint4 lds_raw[];
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(lds_raw);
    Fe<F>* lt = sh + ((size_t)1 << (ns + cbits));  // local twiddles: omega_{2^ns}^j, j < 2^(ns-1)
    const unsigned C = 1u << cbits, tile_n = 1u << (ns + cbits);
    const size_t half_n = (size_t)1 << (log_n - 1);
    // block id -> (H, tbase): the 2^s0 low positions are cut into groups of C
    const size_t groups = ((size_t)1 << s0) >> cbits;  // >= 1
    const size_t H = blockIdx.x / groups, tbase = (blockIdx.x % groups) << cbits;
    for (unsigned j = threadIdx.x; j < (1u << ns) / 2; j += NTT_PASS_BLOCK) lt[j] = tw[(size_t)j << (log_n - ns)];
    for (unsigned e = threadIdx.x; e < tile_n; e += NTT_PASS_BLOCK) {
        unsigned c = e & (C - 1), l = e >> cbits;
        size_t t = tbase + c;
        Fe<F> v = in[(H << (s0 + ns)) | ((size_t)l << s0) | t];
        if (s0) {  // twist by omega_N^(t * rev_ns(l)), N = 2^(s0+ns): index into the omega^i table with the sign fold
            size_t idx = (t * (size_t)(__brev(l) >> (32 - ns))) << (log_n - s0 - ns);
            Fe<F> w = tw[idx & (half_n - 1)];
            if (idx & half_n) w = fe_neg<F>(w);
            v = fe_mul<F>(v, w);
        }
        sh[e] = v;  // e == l * C + c
    }
    __syncthreads();
    for (unsigned st = 0; st < ns; st++) {
        for (unsigned bf = threadIdx.x; bf < tile_n / 2; bf += NTT_PASS_BLOCK) {
            unsigned c = bf & (C - 1), pr = bf >> cbits;
            unsigned lo_bits = pr & ((1u << st) - 1);
            unsigned l0 = ((pr >> st) << (st + 1)) | lo_bits, l1 = l0 | (1u << st);
            Fe<F> w = lt[lo_bits << (ns - st - 1)];
            Fe<F> u = sh[(l0 << cbits) | c], x = fe_mul<F>(sh[(l1 << cbits) | c], w);
            sh[(l0 << cbits) | c] = fe_add<F>(u, x);
            sh[(l1 << cbits) | c] = fe_sub<F>(u, x);
        }
        __syncthreads();
    }
    for (unsigned e = threadIdx.x; e < tile_n; e += NTT_PASS_BLOCK) {
        unsigned c = e & (C - 1), l = e >> cbits;
        Fe<F> v = sh[e];
        if (do_scale) v = fe_mul<F>(v, scale);
        if (out_canonical) v = fe_from_mont<F>(v);
        out[(H << (s0 + ns)) | ((size_t)l << s0) | (tbase + c)] = v;
    }
}


// ---- wave-resident passes (log_n >= 12) ------------------------------------------------------------------------------------
// The north-star shape: butterflies in registers and across lanes, LDS only as the tile that turns strided sub-transforms into
// coalesced rows.  A pass covers stages s0+1 .. s0+ns (ns <= 8) of the same twisted decimation-in-time as ntt_pass_kernel.
// A workgroup of 8 waves owns a tile of 2048 elements = 2^ns rows (index l inside the sub-transform) x 2^(11-ns) neighbouring
// columns; a wave owns 256 of them (slot: l = slot mod 2^ns, column = slot >> ns) as 4 per lane, on the radix-2^29 layer
// (field29.cuh).  Stage st pairs the slots that differ in bit st.  The stages run as RADIX-4 REGISTER ROUNDS (round 4 of the build):
// in round r a lane holds the four slots that differ in bits (2r, 2r + 1), does stage 2r on (e0, e1), (e2, e3) and stage 2r + 1 on
// (e0, e2), (e1, e3) - four butterflies, three twiddles, no lane idle - and between rounds the wave's 256 elements are transposed
// through a wave-private LDS buffer (nine 256-word limb planes, slot index swizzled s ^ (s >> 1) ^ (s >> 2): every round's access
// pattern is bank-conflict free) that aliases the tile.  Before: stages 1-6 paired LANES (__shfl_xor), which cost 12 selects of a
// nine-limb value and 4 exchanges per lane-stage - 673 v_cndmask (4.1 issue cycles each on gfx950) and 216 ds_bpermute (24 cycles of
// the LDS pipeline each) per pass and lane, 10 % of a pass that runs at its instruction-issue bound.  An odd
// ns ends with a radix-2 round on bit ns - 1 (its second free bit is bit 7, a column bit).
// Twiddles of the stages come from a per-workgroup LDS table of omega_{2^ns}^j (radix-2^29 records), the twist between passes
// from the global omega^i table.  The first pass reads the caller's natural-order, canonical input through the bit-reversed
// tile addressing (no separate permutation pass) and converts on the fly; between passes values are stored as the packed
// 256-bit image of the lazily reduced Montgomery-2^261 residue (no conversion products); the last pass multiplies by 1 (or
// n^-1) on the way out, which is also the conversion to canonical bytes.  2^24: three passes of 8 stages.
struct NttConst29 {
    uint32_t in_c[9];   // 2^522 mod p   (canonical -> Montgomery 2^261)
    uint32_t out_c[9];  // 1 or n^-1, plain (Montgomery 2^261 -> canonical, scaled)
};
constexpr int NTT_W_BLOCK = 512;   // 8 waves
constexpr int NTT_W_TILE_LOG = 11;  // 2048 elements per workgroup

template <class F>
__device__ __forceinline__ F29<F> f29_lane_xchg(const F29<F>& v, int mask) {
    F29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_xor(v.l[i], mask);
    return r;
}
template <class F>
__device__ __forceinline__ F29<F> f29_select(bool c, const F29<F>& a, const F29<F>& b) {
    F29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// 4p with its limbs re-balanced like f29_bias (limb_i += 2^30, limb_{i+1} -= 2): subtracting a TIGHT value below 2^256 limb-wise
// never underflows.  The butterflies' subtrahend is always a fresh product (< 2^255 + p), so this small bias replaces the
// general 64p one and a value grows by < 2^256.1 per stage: eight stages stay below 2^259.3 with no reduction in between.
template <class F>
LURK_HD constexpr uint32_t ntt_bias4(int i) {
    uint64_t carry = 0;
    uint32_t limb = 0;
    for (int k = 0; k <= i; k++) {
        uint64_t x = (uint64_t)f29_mod<F>(k) * 4u + carry;
        limb = (uint32_t)(x & F29_MASK);
        carry = x >> 29;
        if (k == 8) limb = (uint32_t)x;
    }
    uint32_t v = limb;
    if (i < 8) v += 1u << 30;
    if (i > 0) v -= 2u;
    return v;
}
// (u, v) <- (u + w v, u - w v): u tight-limbed on entry (value < 2^260), v tight; both results carried (tight limbs), lazily
// reduced: each grows by < 2^256.1
template <class F>
__device__ __forceinline__ void ntt_bfly(F29<F>& u, F29<F>& v, const F29<F>& w) {
    const F29<F> x = f29_mul<F>(v, w);  // tight, < 2^255 + p
    F29<F> d;
#pragma unroll
    for (int i = 0; i < 9; i++) d.l[i] = u.l[i] + (ntt_bias4<F>(i) - x.l[i]);
    v = f29_carry<F>(d);
    u = f29_carry<F>(f29_add<F>(u, x));
}

// zero bits inserted at positions p < q of a 6-bit lane id: the 8-bit slot of the lane's element 0 in the round that pairs bits p and q
__device__ __forceinline__ unsigned ntt_ins2(unsigned lane, unsigned p, unsigned q) {
    const unsigned x = (lane & ((1u << p) - 1u)) | ((lane >> p) << (p + 1));
    return (x & ((1u << q) - 1u)) | ((x >> q) << (q + 1));
}
// position of slot s inside a limb plane of the exchange buffer: with this swizzle the 32 lanes an LDS cycle serves hit 32 different
// banks in every round's layout (checked exhaustively for ns = 5 .. 8)
__device__ __forceinline__ unsigned ntt_swz(unsigned s) { return (s ^ (s >> 1) ^ (s >> 2)) & 255u; }
constexpr unsigned NTT_XCHG_WORDS = 9 * 256;  // per wave

template <class F, bool FIRST>
__global__ __launch_bounds__(NTT_W_BLOCK) void ntt_wave_pass_kernel(const Fe<F>* __restrict__ in, Fe<F>* __restrict__ out, const Fe<F>* __restrict__ tw,
                                                                      unsigned log_n, unsigned s0, unsigned ns, NttConst29 cst, int last) {
    extern __shared__ uint32_t lds_w[];
    const unsigned cbits = NTT_W_TILE_LOG - ns, rows = 1u << ns, cols = 1u << cbits;
    const unsigned row_words = cols * 8 + 2;  // 8 bytes of padding per row: the column reads of a wave spread over all banks
    uint32_t* tile = lds_w;
    const unsigned tile_words = rows * row_words, xchg_words = (NTT_W_BLOCK / 64) * NTT_XCHG_WORDS;
    uint32_t* lt = lds_w + (tile_words > xchg_words ? tile_words : xchg_words);  // omega_{2^ns}^j, j < 2^(ns-1), radix-2^29 records
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t half_n = (size_t)1 << (log_n - 1);
    size_t H = 0, tbase = 0;
    if (FIRST) {
        tbase = (size_t)blockIdx.x << cbits;  // 2^cbits consecutive values of rev(H)
    } else {
        const size_t groups = ((size_t)1 << s0) >> cbits;
        H = blockIdx.x / groups;
        tbase = (blockIdx.x % groups) << cbits;
    }
    for (unsigned j = tid; j < rows / 2; j += NTT_W_BLOCK) {
        const F29<F> w = f29_from_mont256<F>(tw[(size_t)j << (log_n - ns)]);
#pragma unroll
        for (int i = 0; i < 9; i++) lt[j * P29_STRIDE + i] = w.l[i];
    }
    // ---- tile in: rows of 2^cbits neighbouring elements
    for (unsigned e = tid; e < (1u << NTT_W_TILE_LOG); e += NTT_W_BLOCK) {
        const unsigned col = e & (cols - 1), row = e >> cbits;
        // FIRST: row r holds the natural-order elements (r << (log_n - ns)) | rev(H): it is l = rev_ns(r) of sub-transform H
        const size_t src = FIRST ? (((size_t)row << (log_n - ns)) | (tbase + col)) : ((H << (s0 + ns)) | ((size_t)row << s0) | (tbase + col));
        const uint4* g = reinterpret_cast<const uint4*>(in + src);
        const uint4 lo = g[0], hi = g[1];
        uint32_t* d = tile + row * row_words + col * 8;
        d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w; d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
    }
    // ---- the twist factors of the lane's four elements (passes after the first): one random 32-byte gather each from the omega table
    // (HBM in the last pass: 2^23 distinct entries), issued ahead of the barrier so that their latency runs under the wait for the
    // other waves' tile rows and under the LDS reads behind it.  (Ahead of the tile-in they gain nothing: the returns are counted in
    // order, so the tile loop's first wait would wait for the gathers too.)
    const unsigned lmask = rows - 1, wave_cols = 1u << (8 - ns);
    const unsigned base0 = ntt_ins2(lane, 0, 1);
    uint4 twist_lo[4], twist_hi[4];
    bool twist_neg[4];
    if (!FIRST) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned slot = base0 | (unsigned)k, l = slot & lmask, col = wave * wave_cols + (slot >> ns);
            const size_t t = tbase + col;
            const size_t idx = (t * (size_t)(__brev(l) >> (32 - ns))) << (log_n - s0 - ns);
            const uint4* g = reinterpret_cast<const uint4*>(tw + (idx & (half_n - 1)));
            twist_lo[k] = g[0];
            twist_hi[k] = g[1];
            twist_neg[k] = (idx & half_n) != 0;
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the gathers up here: the scheduler otherwise sinks them to their first use
    }
    __syncthreads();
    // ---- registers: 4 elements per lane, in the layout of round 0 (slots that differ in bits 0 and 1)
    F29<F> e[4];
    const unsigned nrounds = (ns + 1) / 2;
    unsigned p = 0, q = 1;  // ns >= 5
    unsigned base = base0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned slot = base | ((k & 1u) << p) | ((unsigned)(k >> 1) << q), l = slot & lmask, col = wave * wave_cols + (slot >> ns);
        const unsigned row = FIRST ? (__brev(l) >> (32 - ns)) : l;
        const uint32_t* sp = tile + row * row_words + col * 8;
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = sp[i];
        if (FIRST) {
            F29<F> c;
#pragma unroll
            for (int i = 0; i < 9; i++) c.l[i] = cst.in_c[i];
            e[k] = f29_mul<F>(f29_from_plain<F>(x), c);
        } else {
            e[k] = f29_from_plain<F>(x);
            Fe<F> w;
            w.l[0] = twist_lo[k].x; w.l[1] = twist_lo[k].y; w.l[2] = twist_lo[k].z; w.l[3] = twist_lo[k].w;
            w.l[4] = twist_hi[k].x; w.l[5] = twist_hi[k].y; w.l[6] = twist_hi[k].z; w.l[7] = twist_hi[k].w;
            if (twist_neg[k]) w = fe_neg<F>(w);
            e[k] = f29_mul<F>(e[k], f29_from_mont256<F>(w));
        }
    }
    __syncthreads();  // every wave has read its elements: the tile's memory now serves as the waves' exchange buffers
    uint32_t* xb = lds_w + wave * NTT_XCHG_WORDS;
    // ---- rounds
#pragma unroll 1
    for (unsigned r = 0; r < nrounds; r++) {
        const bool two = 2 * r + 1 < ns;
        p = 2 * r;
        q = two ? 2 * r + 1 : 7;
        base = ntt_ins2(lane, p, q);
        if (r) {  // the wave's elements back from the exchange buffer, in this round's layout
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned a = ntt_swz(base | ((k & 1u) << p) | ((unsigned)(k >> 1) << q));
#pragma unroll
                for (int i = 0; i < 9; i++) e[k].l[i] = xb[i * 256 + a];
            }
        }
        const unsigned l0 = base & lmask;
        {   // stage p: (e0, e1), (e2, e3); the twiddle depends on the bits of l below p, which the four elements share
            const F29<F> w = ld_const29<F>(lt + (size_t)((l0 & ((1u << p) - 1u)) << (ns - p - 1)) * P29_STRIDE);
            ntt_bfly<F>(e[0], e[1], w);
            ntt_bfly<F>(e[2], e[3], w);
        }
        if (two) {  // stage q = p + 1: (e0, e2) with bit p of l clear, (e1, e3) with it set
            const unsigned j0 = l0 & ((1u << q) - 1u);
            ntt_bfly<F>(e[0], e[2], ld_const29<F>(lt + (size_t)(j0 << (ns - q - 1)) * P29_STRIDE));
            ntt_bfly<F>(e[1], e[3], ld_const29<F>(lt + (size_t)((j0 | (1u << p)) << (ns - q - 1)) * P29_STRIDE));
        }
        if (r + 1 < nrounds) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned a = ntt_swz(base | ((k & 1u) << p) | ((unsigned)(k >> 1) << q));
#pragma unroll
                for (int i = 0; i < 9; i++) xb[i * 256 + a] = e[k].l[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();  // the exchange buffers are done with: the memory is the tile again
    // ---- tile out (each wave rewrites only its own columns), natural l; the lane's slots are those of the last round's layout
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned slot = base | ((k & 1u) << p) | ((unsigned)(k >> 1) << q), l = slot & lmask, col = wave * wave_cols + (slot >> ns);
        uint32_t x[8];
        if (last) {
            F29<F> c;
#pragma unroll
            for (int i = 0; i < 9; i++) c.l[i] = cst.out_c[i];
            const F29<F> u = f29_mul<F>(e[k], c);  // value / 2^261 (times n^-1): tight, < p + 1
            f29_pack<F>(u, x);
            fe_cond_sub<F>(x);
        } else {
            f29_pack<F>(f29_reduce<F>(e[k]), x);  // < 2^255.1: the packed image of the lazy residue
        }
        uint32_t* d = tile + l * row_words + col * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = x[i];
    }
    __syncthreads();
    for (unsigned e2 = tid; e2 < (1u << NTT_W_TILE_LOG); e2 += NTT_W_BLOCK) {
        unsigned col, row;
        size_t dst;
        if (FIRST) {  // column = one contiguous sub-transform of the (bit-reversed order) working array
            row = e2 & lmask;
            col = e2 >> ns;
            const size_t Hc = __brevll((unsigned long long)(tbase + col)) >> (64 - (log_n - ns));
            dst = (Hc << ns) | row;
        } else {
            col = e2 & (cols - 1);
            row = e2 >> cbits;
            dst = (H << (s0 + ns)) | ((size_t)row << s0) | (tbase + col);
        }
        const uint32_t* sp = tile + row * row_words + col * 8;
        uint4* g = reinterpret_cast<uint4*>(out + dst);
        g[0] = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        g[1] = make_uint4(sp[4], sp[5], sp[6], sp[7]);
    }
}
static size_t ntt_wave_lds(unsigned ns) {
    const size_t rows = (size_t)1 << ns, cols = (size_t)1 << (NTT_W_TILE_LOG - ns);
    const size_t tile = rows * (cols * 8 + 2), xchg = (size_t)(NTT_W_BLOCK / 64) * NTT_XCHG_WORDS;  // the exchange buffers alias the tile
    return ((tile > xchg ? tile : xchg) + (rows / 2 + 1) * P29_STRIDE) * 4;
}

template <class F>
static Fe<F> host_omega(unsigned log_n, bool inverse) {
    Fe<F> w;
    for (int i = 0; i < 8; i++) w.l[i] = F::w32(i);
    for (unsigned i = log_n; i < 32; i++) w = fe_sqr<F>(w);
    if (inverse) w = fe_inv<F>(w);
    return w;
}

// Twiddle tables and the bit-reversal scratch buffer are cached per (device, field, log_n, direction):
// generating n/2 powers costs more field multiplications than the transform itself.
struct NttPlan {
    DevBuf tw;
};
static std::mutex g_plan_mu;
static std::map<std::tuple<int, int, unsigned, bool>, std::unique_ptr<NttPlan>> g_plans;

template <class F>
static NttPlan& ntt_plan(unsigned log_n, bool inverse, hipStream_t s) {
    int dev = 0;
    LURK_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto key = std::make_tuple(dev, (int)F::ID, log_n, inverse);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        const size_t n = (size_t)1 << log_n;
        auto plan = std::make_unique<NttPlan>();
        plan->tw.alloc((n / 2 ? n / 2 : 1) * 32);
        if (n > 1) {
            Fe<F> omega = host_omega<F>(log_n, inverse);
            hipLaunchKernelGGL((ntt_twiddle_kernel<F>), dim3(div_up(n / 2, 256)), dim3(256), 0, s, omega, n / 2, plan->tw.template as<Fe<F>>());
            LURK_HIP_CHECK(hipGetLastError());
            LURK_HIP_CHECK(hipStreamSynchronize(s));
        }
        it = g_plans.emplace(key, std::move(plan)).first;
    }
    return *it->second;
}

template <class F>
static void ntt_device(void* d_data, unsigned log_n, bool inverse, hipStream_t s) {
    LURK_REQUIRE(log_n <= 28, "log_n too large");
    const size_t n = (size_t)1 << log_n;
    NttPlan& plan = ntt_plan<F>(log_n, inverse, s);
    // the ping-pong scratch is stream-ordered (allocated and freed on `s`): concurrent transforms on one plan never
    // share it and the call does not synchronise
    Fe<F>* tmp = nullptr;
    LURK_HIP_CHECK(hipMallocAsync((void**)&tmp, n * 32, s));
    Fe<F>* data = (Fe<F>*)d_data;
    const Fe<F>* tw = plan.tw.template as<Fe<F>>();
    allow_dynamic_lds((const void*)ntt_pass_kernel<F>, 160 * 1024);
    ProfScope ps("ntt", s);
    if (log_n >= 12) {
        NttConst29 cst;
        {
            Fe<F> v = fe_one<F>();  // 2^256 mod p as a plain integer -> 2^522 mod p
            for (int d = 0; d < 522 - 256; d++) v = fe_add<F>(v, v);
            const F29<F> a = f29_from_plain<F>(v.l);
            Fe<F> o = fe_from_mont<F>(inverse ? fe_inv<F>(fe_from_u64<F>((uint64_t)n)) : fe_one<F>());  // plain n^-1 or 1
            const F29<F> b = f29_from_plain<F>(o.l);
            for (int i = 0; i < 9; i++) { cst.in_c[i] = a.l[i]; cst.out_c[i] = b.l[i]; }
        }
        const unsigned passes = (log_n + 7) / 8;
        unsigned s0 = 0;
        for (unsigned p = 0; p < passes; p++) {
            const unsigned ns = (log_n - s0 + (passes - p) - 1) / (passes - p);  // as even as possible, <= 8
            const bool first = p == 0, last = p + 1 == passes;
            const size_t lds = ntt_wave_lds(ns);
            const Fe<F>* src = first ? data : tmp;
            Fe<F>* dst = last ? data : tmp;
            // per-pass HIP-event times beside the whole transform's ("ntt"): what bench.py quotes next to the rocprofv3 summary
            static const char* const pass_names[4] = {"pass_ntt_0", "pass_ntt_1", "pass_ntt_2", "pass_ntt_3"};
            ProfScope pp(pass_names[p < 4 ? p : 3], s);
            if (first) {
                allow_dynamic_lds((const void*)ntt_wave_pass_kernel<F, true>, 160 * 1024);
                hipLaunchKernelGGL((ntt_wave_pass_kernel<F, true>), dim3((unsigned)(n >> NTT_W_TILE_LOG)), dim3(NTT_W_BLOCK), lds, s, src, tmp, tw, log_n, 0u, ns,
                                   cst, 0);
            } else {
                allow_dynamic_lds((const void*)ntt_wave_pass_kernel<F, false>, 160 * 1024);
                hipLaunchKernelGGL((ntt_wave_pass_kernel<F, false>), dim3((unsigned)(n >> NTT_W_TILE_LOG)), dim3(NTT_W_BLOCK), lds, s, src, dst, tw, log_n, s0, ns,
                                   cst, last ? 1 : 0);
            }
            s0 += ns;
        }
        hipError_t launch_err = hipGetLastError();
        (void)hipFreeAsync(tmp, s);
        LURK_HIP_CHECK(launch_err);
        return;
    }
    if (log_n >= 10) hipLaunchKernelGGL((ntt_bitrev_tiled_kernel<F>), dim3((unsigned)(n >> 10)), dim3(256), 0, s, data, tmp, log_n);
    else hipLaunchKernelGGL((ntt_bitrev_kernel<F>), dim3(div_up(n, 256)), dim3(256), 0, s, data, tmp, log_n);
    Fe<F> scale = fe_one<F>();
    if (inverse) scale = fe_inv<F>(fe_from_u64<F>((uint64_t)n));
    // pass plan: first pass as deep as the tile allows, the rest in passes of <= 9 stages (C >= 4: rows of >= 128 B)
    unsigned s0 = 0;
    bool first = true;
    do {
        unsigned left = log_n - s0;
        unsigned ns = first ? (left < (unsigned)NTT_TILE_LOG ? left : NTT_TILE_LOG) : (left <= 9 ? left : ((left + 1) / 2 < 8 ? (left + 1) / 2 : 8));
        unsigned cbits = first ? 0 : (NTT_TILE_LOG - ns < s0 ? NTT_TILE_LOG - ns : s0);
        bool last = s0 + ns == log_n;
        size_t lds = (((size_t)1 << (ns + cbits)) + ((size_t)1 << ns) / 2 + 1) * sizeof(Fe<F>);
        hipLaunchKernelGGL((ntt_pass_kernel<F>), dim3((unsigned)(n >> (ns + cbits))), dim3(NTT_PASS_BLOCK), lds, s, tmp, last ? data : tmp, tw,
                           log_n, s0, ns, cbits, scale, (last && inverse) ? 1 : 0, last ? 1 : 0);
        s0 += ns;
        first = false;
    } while (s0 < log_n);
    hipError_t launch_err = hipGetLastError();
    (void)hipFreeAsync(tmp, s);
    LURK_HIP_CHECK(launch_err);
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_ntt_dev(int field_id, void* d_inout, unsigned log_n, int inverse, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id == 0 || field_id == 1, "NTT is offered over the Pasta fields only");
        LURK_REQUIRE(d_inout, "null buffer");
        if (field_id == 0) ntt_device<PallasFp>(d_inout, log_n, inverse != 0, (hipStream_t)stream);
        else ntt_device<PallasFq>(d_inout, log_n, inverse != 0, (hipStream_t)stream);
    });
}
int lurk_hip_ntt(int field_id, void* inout, unsigned log_n, int inverse) {
    return guarded([&] {
        LURK_REQUIRE(field_id == 0 || field_id == 1, "NTT is offered over the Pasta fields only");
        LURK_REQUIRE(inout, "null buffer");
        LURK_REQUIRE(log_n <= 28, "log_n too large");
        size_t n = (size_t)1 << log_n;
        DevBuf d(n * 32);
        LURK_HIP_CHECK(hipMemcpy(d.p, inout, n * 32, hipMemcpyHostToDevice));
        if (field_id == 0) ntt_device<PallasFp>(d.p, log_n, inverse != 0, nullptr);
        else ntt_device<PallasFq>(d.p, log_n, inverse != 0, nullptr);
        LURK_HIP_CHECK(hipStreamSynchronize(nullptr));
        LURK_HIP_CHECK(hipMemcpy(inout, d.p, n * 32, hipMemcpyDeviceToHost));
    });
}
}

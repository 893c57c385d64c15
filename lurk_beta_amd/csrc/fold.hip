// fold.hip - the device-resident part of one relaxed-R1CS folding step.
//
// Replaces, for the arithmetic only, what arecibo's NIFS::prove does on the CPU between the two commitments
// of a step (caller: RecursiveSNARK::prove_step, /root/reference/src/proof/nova.rs:291-293; arecibo itself is
// an un-vendored git dependency, /root/reference/Cargo.toml:128, so this follows the published Nova
// construction and oracle/oracle.c restates it - parity unpinned, no golden vectors upstream):
//   R1CSShape::multiply_vec   (A z, B z, C z)                          -> lurk_hip_r1cs_multiply_vec_dev
//   R1CSShape::commit_T       T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1 -> lurk_hip_r1cs_cross_term_dev
//   RelaxedR1CSWitness::fold  W <- W1 + r W2,  E <- E1 + r T            -> lurk_hip_fold_vec_dev
// With these the witness vectors never leave HBM between commit(W), commit(T) and the next step.
//
// Layout.  A shape keeps its three CSR matrices resident: row pointers (u32), and per non-zero ONE 8-byte
// record {column, coefficient id}.  R1CS coefficients repeat massively (+-1, small constants), so the 32-byte
// values are replaced by ids into a dictionary of distinct coefficients built once at creation (radix-2^29
// limbs, 48-byte records, read through L2): 8 B per non-zero instead of 36 B.  z = [W | u | X] as arecibo lays
// it out; all field elements cross the ABI as 32-byte Montgomery(2^256) values.
//
// Kernels: one lane per row (the step circuit's rows hold 3-4 entries; neighbouring lanes read neighbouring
// CSR records); the few long rows (bit decompositions: ~255 entries) go to a second launch with 16 lanes per row.  A row's inner product is accumulated in 17 unreduced 64-bit columns (poseidon29.cuh: Dot29)
// and reduced once.  The cross-term kernel runs the six inner products of a row (A, B, C times z1, z2) from one
// pass over the matrices and finishes T as a four-term lazy row: AZ1*BZ2 + AZ2*BZ1 + (-u1)*CZ2 + (-u2)*CZ1.
// Both kernels are bound by the random 32-byte gathers from z (HBM / L2), not by arithmetic.
#include <memory>
#include <unordered_map>

#include "common.hpp"
#include "poseidon29.cuh"

namespace lurk {

constexpr int FOLD_BLOCK = 256;

// The cross term and the two folds are links of the step's serial chain (cross term -> commit(T) -> r -> folds -> next cross term) and
// are bound by HBM latency, not by issue slots; what shares the device with them is an accumulation of the NEXT step's commit(W2)
// (step.hip: LURK_MSM_SUBMIT_FOLLOW), whose waves are older and - at equal priority - served first by the instruction arbiter:
// measured round 6, a 60 us fold took 320 us and the cross term 590 instead of 410.  Raised wave priority lets these waves issue when
// their loads come back (3: the class of commit(T)'s own short kernels; a followed commitment's tail runs at 2, accumulations at 0-1).
__device__ __forceinline__ void fold_wave_prio() { __builtin_amdgcn_s_setprio(3); }

struct CsrDev {
    DevBuf rowptr;  // u32 x (rows + 1)
    DevBuf ent;     // uint2 {col, coefficient id} x nnz
    size_t nnz = 0;
};

struct R1csShape {
    int field_id = 0;
    size_t num_cons = 0, num_vars = 0, num_io = 0;
    CsrDev m[3];
    DevBuf dict;  // distinct coefficients, P29_STRIDE words each (canonical Montgomery-2^261 limbs)
    size_t dict_size = 0;
    DevBuf long_rows;  // u32 row ids with more than FOLD_LONG entries in A, B or C
    size_t n_long = 0;
    int device = 0;
};

// ---- rows --------------------------------------------------------------------------------------------------
// Lazy accumulator of one row value: terms a*b (both tight) are added as unreduced 17-column products.
// 45 products per column fit 64 bits: the columns are normalised every fourth term.  Every term adds < 2^253
// to the value: every 64 terms the reduced partial sum re-enters as one term (times the Montgomery one) so that
// rows of any length stay below 2^261.
template <class P>
struct RowAcc {
    Dot29<P> acc;
    uint32_t since, terms;
};
template <class P>
__device__ __forceinline__ void row_init(RowAcc<P>& r) {
    dot29_init<P>(r.acc);
    r.since = 0;
    r.terms = 0;
}
template <class P>
__device__ __forceinline__ void row_mac(RowAcc<P>& r, const F29<P>& a, const F29<P>& b, const uint32_t* one29) {
    if (r.terms == 64) {
        F29<P> part = dot29_finish<P>(r.acc);
        dot29_init<P>(r.acc);
        dot29_mac<P>(r.acc, part, ld_const29<P>(one29));
        r.terms = 1;
        r.since = 1;
    }
    if (r.since == 4) {
        dot29_carry<P>(r.acc);
        r.since = 0;
    }
    dot29_mac<P>(r.acc, a, b);
    r.since++;
    r.terms++;
}

constexpr uint32_t FOLD_LONG = 32;  // rows with more entries (in any of A, B, C) go to the wave-per-row kernel
// Measured in isolation at rc = 100 (bench_tools/fold_bench.py, round 6): a batch of 4 entries keeps 180-200 registers live (two waves
// per SIMD), a batch of 2 keeps 144 (three waves): 0.227 -> 0.200 ms for the cached-products kernel, 0.323 -> 0.312 for the six-gather
// one - the kernels are bound by gather latency that only more resident waves hide (1.6 waves per SIMD on average in round 4's counters).
#ifndef LURK_FOLD_BATCH
#define LURK_FOLD_BATCH 2
#endif
#ifndef LURK_FOLD_MIN_WAVES
#define LURK_FOLD_MIN_WAVES 1  // (HIP: minimum waves per SIMD the cross-term kernels are compiled for)
#endif
constexpr int FOLD_BATCH = LURK_FOLD_BATCH;  // entries whose loads are issued together by one lane

struct CsrView {
    const uint32_t* rowptr;
    const uint2* ent;
};

// one lane, one (short) row of one matrix, NV vectors: entries are fetched FOLD_BATCH at a time so that the
// dependent loads (record -> coefficient, z values) of a batch are in flight together
template <class P, int NV>
__device__ __forceinline__ void fold_row_lane(const CsrView& m, const uint32_t* __restrict__ dict, const uint32_t* __restrict__ one29,
                                              uint32_t lo, uint32_t hi, const Fe<P>* const* z, F29<P>* out) {
    RowAcc<P> acc[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) row_init<P>(acc[v]);
    for (uint32_t k = lo; k < hi; k += FOLD_BATCH) {
        uint2 e[FOLD_BATCH];
#pragma unroll
        for (int u = 0; u < FOLD_BATCH; u++) e[u] = k + u < hi ? m.ent[k + u] : make_uint2(0u, 0u);
        F29<P> c[FOLD_BATCH];
        Fe<P> zz[FOLD_BATCH][NV];
#pragma unroll
        for (int u = 0; u < FOLD_BATCH; u++) {
            c[u] = ld_const29<P>(dict + (size_t)e[u].y * P29_STRIDE);
#pragma unroll
            for (int v = 0; v < NV; v++) zz[u][v] = z[v][e[u].x];
        }
#pragma unroll
        for (int u = 0; u < FOLD_BATCH; u++)
            if (k + u < hi) {
#pragma unroll
                for (int v = 0; v < NV; v++) row_mac<P>(acc[v], c[u], f29_from_mont256<P>(zz[u][v]), one29);
            }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) out[v] = dot29_finish<P>(acc[v].acc);
}

// FOLD_GROUP lanes, one long row of one matrix (64 / FOLD_GROUP rows per wave): lane g of the group runs a RowAcc
// over the entries lo + g, lo + g + FOLD_GROUP, ... and reduces it; the reduced values are then summed across
// the group - limb-wise (two butterfly steps between carry passes: 4 x 2^29 < 2^32) while the row is short enough
// for the sum to stay below 2^261 (<= 256 entries: < 256 * 2^252 + 16 p), otherwise one by one into the group
// leader's accumulator (times the Montgomery one).  Result valid on the group leader.
constexpr int FOLD_GROUP = 16;

template <class P, int NV>
__device__ __forceinline__ void fold_row_wave(const CsrView& m, const uint32_t* __restrict__ dict, const uint32_t* __restrict__ one29,
                                              uint32_t lo, uint32_t hi, const Fe<P>* const* z, F29<P>* out) {
    static_assert(NV == 1, "one vector per pass");
    const uint32_t gl = threadIdx.x & (FOLD_GROUP - 1);
    RowAcc<P> acc;
    row_init<P>(acc);
    for (uint32_t k = lo + gl; k < hi; k += FOLD_GROUP * FOLD_BATCH) {  // FOLD_BATCH entries' loads in flight together
        uint2 e[FOLD_BATCH];
#pragma unroll
        for (int u = 0; u < FOLD_BATCH; u++) e[u] = k + u * FOLD_GROUP < hi ? m.ent[k + u * FOLD_GROUP] : make_uint2(0u, 0u);
        F29<P> c[FOLD_BATCH];
        Fe<P> zz[FOLD_BATCH];
#pragma unroll
        for (int u = 0; u < FOLD_BATCH; u++) {
            c[u] = ld_const29<P>(dict + (size_t)e[u].y * P29_STRIDE);
            zz[u] = z[0][e[u].x];
        }
#pragma unroll
        for (int u = 0; u < FOLD_BATCH; u++)
            if (k + u * FOLD_GROUP < hi) row_mac<P>(acc, c[u], f29_from_mont256<P>(zz[u]), one29);
    }
    F29<P> part = dot29_finish<P>(acc.acc);  // tight, < 2^258.1
    if (hi - lo <= 256) {
#pragma unroll
        for (int off = FOLD_GROUP / 2; off >= 1; off >>= 1) {
#pragma unroll
            for (int i = 0; i < 9; i++) part.l[i] += __shfl_down(part.l[i], off);
            if (off == FOLD_GROUP / 4 || off == 1) part = f29_carry<P>(part);
        }
        out[0] = part;
    } else {
        RowAcc<P> tot;
        row_init<P>(tot);
        const int leader = (threadIdx.x & 63) & ~(FOLD_GROUP - 1);
#pragma unroll 1
        for (int g = 0; g < FOLD_GROUP; g++) {
            F29<P> v;
#pragma unroll
            for (int i = 0; i < 9; i++) v.l[i] = __shfl(part.l[i], leader + g);
            row_mac<P>(tot, v, ld_const29<P>(one29), one29);
        }
        out[0] = dot29_finish<P>(tot.acc);
    }
}

template <class P>
__device__ __forceinline__ void fold_store(Fe<P>* dst, const F29<P>& v) {
    *dst = f29_to_mont256<P>(v);
}

struct R1csDev {
    CsrView a, b, c;
    const uint32_t* dict;
    size_t dict_size;
    size_t rows;
    const uint32_t* long_rows;  // rows with > FOLD_LONG entries in A, B or C
    uint32_t n_long;
};

__device__ __forceinline__ bool fold_is_long(const R1csDev& s, size_t row, uint32_t* lo, uint32_t* hi) {
    lo[0] = s.a.rowptr[row]; hi[0] = s.a.rowptr[row + 1];
    lo[1] = s.b.rowptr[row]; hi[1] = s.b.rowptr[row + 1];
    lo[2] = s.c.rowptr[row]; hi[2] = s.c.rowptr[row + 1];
    return hi[0] - lo[0] > FOLD_LONG || hi[1] - lo[1] > FOLD_LONG || hi[2] - lo[2] > FOLD_LONG;
}

// Workgroup -> row block, XCD-aware.  Consecutive workgroup ids go round the 8 XCDs (each with its own 4 MiB L2), and the rows of the
// step circuit are frame-structured: a frame's rows gather almost only that frame's ~9 000 columns of z (multiframe.rs:699-702).  With
// the identity mapping every XCD's L2 sees every frame's columns (each 32-byte element of z is fetched into up to 8 L2s); here XCD x
// walks ONE contiguous eighth of the rows, so a frame's columns of z1 / z2 are fetched by one L2 and hit there for the rest of the
// frame's rows.
constexpr unsigned FOLD_XCDS = 8;
__device__ __forceinline__ size_t fold_row_block(unsigned b, unsigned nblocks) {
    const unsigned per = (nblocks + FOLD_XCDS - 1) / FOLD_XCDS;
    return (size_t)(b % FOLD_XCDS) * per + b / FOLD_XCDS;
}

// LONG = false: one lane per row, long rows skipped;  LONG = true: FOLD_GROUP lanes per entry of long_rows
template <class P, bool LONG>
__global__ __launch_bounds__(FOLD_BLOCK) void r1cs_multiply_vec_kernel(R1csDev s, const Fe<P>* __restrict__ z, Fe<P>* __restrict__ az,
                                                                         Fe<P>* __restrict__ bz, Fe<P>* __restrict__ cz) {
    const uint32_t* one29 = s.dict + s.dict_size * P29_STRIDE;  // the Montgomery one closes the dictionary
    const Fe<P>* zs[1] = {z};
    uint32_t lo[3], hi[3];
    F29<P> r;
    if (!LONG) {
        size_t row = fold_row_block(blockIdx.x, gridDim.x) * FOLD_BLOCK + threadIdx.x;
        if (row >= s.rows || fold_is_long(s, row, lo, hi)) return;
        fold_row_lane<P, 1>(s.a, s.dict, one29, lo[0], hi[0], zs, &r);
        fold_store<P>(az + row, r);
        fold_row_lane<P, 1>(s.b, s.dict, one29, lo[1], hi[1], zs, &r);
        fold_store<P>(bz + row, r);
        fold_row_lane<P, 1>(s.c, s.dict, one29, lo[2], hi[2], zs, &r);
        fold_store<P>(cz + row, r);
    } else {
        size_t w = ((size_t)blockIdx.x * FOLD_BLOCK + threadIdx.x) / FOLD_GROUP;
        const bool live = w < s.n_long;  // a group without a row shadows the last one (the shuffles need every lane)
        size_t row = s.long_rows[live ? w : s.n_long - 1];
        fold_is_long(s, row, lo, hi);
        const bool lead = live && (threadIdx.x & (FOLD_GROUP - 1)) == 0;
        fold_row_wave<P, 1>(s.a, s.dict, one29, lo[0], hi[0], zs, &r);
        if (lead) fold_store<P>(az + row, r);
        fold_row_wave<P, 1>(s.b, s.dict, one29, lo[1], hi[1], zs, &r);
        if (lead) fold_store<P>(bz + row, r);
        fold_row_wave<P, 1>(s.c, s.dict, one29, lo[2], hi[2], zs, &r);
        if (lead) fold_store<P>(cz + row, r);
    }
}

// -u1, -u2 (u_i = z_i[num_vars]) are negated and converted by every lane that finishes a row: two broadcast loads and two
// limb repackings, so that a call keeps no state outside its arguments (concurrent calls on one shape are independent)
// bid / nb: this workgroup's index among the nb workgroups of its kind (the long-row and the short-row blocks of ONE launch)
template <class P, bool LONG>
__device__ __forceinline__ void r1cs_cross_term_body(const R1csDev& s, const Fe<P>* __restrict__ z1, const Fe<P>* __restrict__ z2, size_t u_index,
                                                     Fe<P>* __restrict__ t, unsigned bid, unsigned nb) {
    const uint32_t* one29 = s.dict + s.dict_size * P29_STRIDE;
    const Fe<P>* zs[2] = {z1, z2};
    uint32_t lo[3], hi[3];
    F29<P> a[2], b[2], c[2];
    size_t row;
    Dot29<P> acc;
    dot29_init<P>(acc);
    if (!LONG) {
        row = fold_row_block(bid, nb) * FOLD_BLOCK + threadIdx.x;
        if (row >= s.rows || fold_is_long(s, row, lo, hi)) return;
        // one vector at a time (a second pass re-reads the 8-byte records from L2 but halves the live accumulators:
        // 3 waves/SIMD instead of 2 with spills); T is built up as the row values arrive
#pragma unroll
        for (int v = 0; v < 2; v++) {
            fold_row_lane<P, 1>(s.a, s.dict, one29, lo[0], hi[0], zs + v, a + v);
            fold_row_lane<P, 1>(s.b, s.dict, one29, lo[1], hi[1], zs + v, b + v);
        }
        dot29_mac<P>(acc, a[0], b[1]);
        dot29_mac<P>(acc, a[1], b[0]);
#pragma unroll
        for (int v = 0; v < 2; v++) fold_row_lane<P, 1>(s.c, s.dict, one29, lo[2], hi[2], zs + v, c + v);
    } else {
        size_t w = ((size_t)bid * FOLD_BLOCK + threadIdx.x) / FOLD_GROUP;
        const bool live = w < s.n_long;  // a group without a row shadows the last one (the shuffles need every lane)
        row = s.long_rows[live ? w : s.n_long - 1];
        fold_is_long(s, row, lo, hi);
#pragma unroll
        for (int v = 0; v < 2; v++) {
            fold_row_wave<P, 1>(s.a, s.dict, one29, lo[0], hi[0], zs + v, a + v);
            fold_row_wave<P, 1>(s.b, s.dict, one29, lo[1], hi[1], zs + v, b + v);
            fold_row_wave<P, 1>(s.c, s.dict, one29, lo[2], hi[2], zs + v, c + v);
        }
        if (!live || (threadIdx.x & (FOLD_GROUP - 1))) return;
        dot29_mac<P>(acc, a[0], b[1]);
        dot29_mac<P>(acc, a[1], b[0]);
    }
    dot29_mac<P>(acc, f29_from_mont256<P>(fe_neg<P>(z1[u_index])), c[1]);  // 32 (-u) 2^256: lazy Montgomery-2^261 form, tight, < 2^259
    dot29_mac<P>(acc, f29_from_mont256<P>(fe_neg<P>(z2[u_index])), c[0]);
    fold_store<P>(t + row, dot29_finish<P>(acc));
}
// ONE launch: the first long_blocks workgroups take the few long rows (bit decompositions: ~255 entries, FOLD_GROUP lanes per row - a
// latency-bound handful of waves), the rest one short row per lane.  As two launches on one stream the long-row kernel ran ALONE first:
// 0.14 ms of a 0.42 ms cross term at rc = 100 with the device nearly idle (profiles/r04_step_timeline_rc100.txt), and commit(T) - which
// waits for T - started that much later.
template <class P>
__global__ __launch_bounds__(FOLD_BLOCK, LURK_FOLD_MIN_WAVES) void r1cs_cross_term_kernel(R1csDev s, const Fe<P>* __restrict__ z1, const Fe<P>* __restrict__ z2,
                                                                       size_t u_index, Fe<P>* __restrict__ t, unsigned long_blocks) {
    fold_wave_prio();
    if (blockIdx.x < long_blocks) r1cs_cross_term_body<P, true>(s, z1, z2, u_index, t, blockIdx.x, long_blocks);
    else r1cs_cross_term_body<P, false>(s, z1, z2, u_index, t, blockIdx.x - long_blocks, gridDim.x - long_blocks);
}

// ---- the cross term with the running products cached (round 6) ---------------------------------------------------------------------
// A z1, B z1, C z1 fold linearly with the running instance (A (z1 + r z2) = A z1 + r A z2), so a folding context keeps them resident
// beside z1 and E (step.hip) and a step gathers from z2 ALONE: three inner products per row instead of six, half the dependent
// 32-byte gather chains and half the Montgomery conversions of r1cs_cross_term_kernel - the kernel is bound by exactly those (L1 tag
// path 55 %, VALU 47 %: profiles/r04_cross_term_pmc.txt).  The cached rows arrive as three coalesced 32-byte reads per lane; A z2, B z2,
// C z2 leave the same way for finish(r), which folds them into the cache in the launch that folds z and E (fold_vecs_kernel).
// The fold of the cache itself rides in the same kernel (prev != nullptr): the products left by the PREVIOUS step and that step's
// challenge, cached row <- cached row + r_prev * previous row, stored back in place and used at once - the step's finish(r) then has
// nothing on the chain between the transcript and the next cross term (the folds of z and E, which the cross term does not read, run on
// a side stream: step.hip), and the three cached vectors are read once per step instead of twice.
template <class P>
struct CachedProducts {
    Fe<P>* a1;  // A z1, B z1, C z1: read, and written back when prev_* are given
    Fe<P>* b1;
    Fe<P>* c1;
    const Fe<P>* prev_a;  // A z2, B z2, C z2 of the previous step (or nullptr: nothing to fold in)
    const Fe<P>* prev_b;
    const Fe<P>* prev_c;
    Fe<P> r_prev;  // the previous step's challenge (Montgomery)
    Fe<P> u1;      // the running u AFTER that fold (Montgomery): the host folds scalars itself
};
template <class P>
__device__ __forceinline__ Fe<P> cached_row(Fe<P>* cur, const Fe<P>* prev, const Fe<P>& r, size_t row) {
    Fe<P> v = cur[row];
    if (prev) {
        v = fe_add<P>(v, fe_mul<P>(r, prev[row]));
        cur[row] = v;
    }
    return v;
}
template <class P, bool LONG>
__device__ __forceinline__ void r1cs_cross_term_cached_body(const R1csDev& s, const Fe<P>* __restrict__ z2, const CachedProducts<P>& cp, size_t u_index,
                                                            Fe<P>* __restrict__ t, Fe<P>* __restrict__ az2, Fe<P>* __restrict__ bz2,
                                                            Fe<P>* __restrict__ cz2, unsigned bid, unsigned nb) {
    const uint32_t* one29 = s.dict + s.dict_size * P29_STRIDE;
    const Fe<P>* zs[1] = {z2};
    uint32_t lo[3], hi[3];
    size_t row;
    bool store = true;
    if (!LONG) {
        row = fold_row_block(bid, nb) * FOLD_BLOCK + threadIdx.x;
        if (row >= s.rows || fold_is_long(s, row, lo, hi)) return;
    } else {
        size_t w = ((size_t)bid * FOLD_BLOCK + threadIdx.x) / FOLD_GROUP;
        const bool live = w < s.n_long;  // a group without a row shadows the last one (the shuffles need every lane)
        row = s.long_rows[live ? w : s.n_long - 1];
        fold_is_long(s, row, lo, hi);
        store = live && (threadIdx.x & (FOLD_GROUP - 1)) == 0;
    }
    // One product at a time, each folded into T's four-term lazy row and stored as soon as it is complete: only ONE row value is live
    // beside the accumulator (three of them at once cost the kernel its third wave per SIMD: 200 registers against 168).
    Dot29<P> acc;
    dot29_init<P>(acc);
    F29<P> v;
    if (!LONG) fold_row_lane<P, 1>(s.a, s.dict, one29, lo[0], hi[0], zs, &v);
    else fold_row_wave<P, 1>(s.a, s.dict, one29, lo[0], hi[0], zs, &v);
    if (store) {
        dot29_mac<P>(acc, v, f29_from_mont256<P>(cached_row<P>(cp.b1, cp.prev_b, cp.r_prev, row)));  // A z2 o B z1
        fold_store<P>(az2 + row, v);
    }
    if (!LONG) fold_row_lane<P, 1>(s.b, s.dict, one29, lo[1], hi[1], zs, &v);
    else fold_row_wave<P, 1>(s.b, s.dict, one29, lo[1], hi[1], zs, &v);
    if (store) {
        dot29_mac<P>(acc, f29_from_mont256<P>(cached_row<P>(cp.a1, cp.prev_a, cp.r_prev, row)), v);  // A z1 o B z2
        fold_store<P>(bz2 + row, v);
    }
    if (!LONG) fold_row_lane<P, 1>(s.c, s.dict, one29, lo[2], hi[2], zs, &v);
    else fold_row_wave<P, 1>(s.c, s.dict, one29, lo[2], hi[2], zs, &v);
    if (!store) return;
    dot29_mac<P>(acc, f29_from_mont256<P>(fe_neg<P>(cp.u1)), v);                                                                          // - u1 C z2
    dot29_mac<P>(acc, f29_from_mont256<P>(fe_neg<P>(z2[u_index])), f29_from_mont256<P>(cached_row<P>(cp.c1, cp.prev_c, cp.r_prev, row)));  // - u2 C z1
    fold_store<P>(cz2 + row, v);
    fold_store<P>(t + row, dot29_finish<P>(acc));
}
template <class P>
__global__ __launch_bounds__(FOLD_BLOCK, LURK_FOLD_MIN_WAVES) void r1cs_cross_term_cached_kernel(R1csDev s, const Fe<P>* __restrict__ z2, CachedProducts<P> cp,
                                                                              size_t u_index, Fe<P>* __restrict__ t, Fe<P>* __restrict__ az2,
                                                                              Fe<P>* __restrict__ bz2, Fe<P>* __restrict__ cz2, unsigned long_blocks) {
    fold_wave_prio();
    if (blockIdx.x < long_blocks) r1cs_cross_term_cached_body<P, true>(s, z2, cp, u_index, t, az2, bz2, cz2, blockIdx.x, long_blocks);
    else r1cs_cross_term_cached_body<P, false>(s, z2, cp, u_index, t, az2, bz2, cz2, blockIdx.x - long_blocks, gridDim.x - long_blocks);
}

// out = a + r b (r: Montgomery 2^256, broadcast).  A pure 96 B / element stream (two reads, one write): a lane takes FOLD_VEC_E
// elements FOLD_BLOCK apart, issues all of their 128-bit loads before the first product (8 loads in flight per lane; with one
// element per trip of a grid-stride loop the kernel sat at 45-53 % of HBM, waiting on each pair of loads in turn), and the grid
// covers the vector exactly once.
constexpr int FOLD_VEC_E = 2;
template <class P>
__global__ __launch_bounds__(FOLD_BLOCK) void fold_vec_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, Fe<P> r, size_t n,
                                                                uint4* __restrict__ out) {
    fold_wave_prio();
    const size_t base = (size_t)blockIdx.x * (FOLD_BLOCK * FOLD_VEC_E) + threadIdx.x;
    uint4 al[FOLD_VEC_E], ah[FOLD_VEC_E], bl[FOLD_VEC_E], bh[FOLD_VEC_E];
#pragma unroll
    for (int e = 0; e < FOLD_VEC_E; e++) {
        const size_t i = base + (size_t)e * FOLD_BLOCK;
        if (i < n) {
            al[e] = a[2 * i];
            ah[e] = a[2 * i + 1];
            bl[e] = b[2 * i];
            bh[e] = b[2 * i + 1];
        }
    }
#pragma unroll
    for (int e = 0; e < FOLD_VEC_E; e++) {
        const size_t i = base + (size_t)e * FOLD_BLOCK;
        if (i >= n) continue;
        Fe<P> x, y;
        x.l[0] = al[e].x; x.l[1] = al[e].y; x.l[2] = al[e].z; x.l[3] = al[e].w; x.l[4] = ah[e].x; x.l[5] = ah[e].y; x.l[6] = ah[e].z; x.l[7] = ah[e].w;
        y.l[0] = bl[e].x; y.l[1] = bl[e].y; y.l[2] = bl[e].z; y.l[3] = bl[e].w; y.l[4] = bh[e].x; y.l[5] = bh[e].y; y.l[6] = bh[e].z; y.l[7] = bh[e].w;
        x = fe_add<P>(x, fe_mul<P>(r, y));
        out[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
        out[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    }
}

// Up to FOLD_VECS_MAX folds out_k = a_k + r b_k in ONE launch (the step's finish(r): [W | u | X], E and the three cached products): a
// workgroup belongs to one vector (first_block[k] <= blockIdx.x < first_block[k + 1]) and does what fold_vec_kernel does for it.  Five
// launches of 30 MB each are each bound by their own ramp-up (fold_vec at rc = 100: 49 % of HBM); one launch of 150 MB is not.
constexpr int FOLD_VECS_MAX = 8;
struct FoldVecs {
    const uint4* a[FOLD_VECS_MAX];
    const uint4* b[FOLD_VECS_MAX];
    uint4* out[FOLD_VECS_MAX];
    size_t n[FOLD_VECS_MAX];
    unsigned first_block[FOLD_VECS_MAX + 1];
    int count;
};
template <class P>
__global__ __launch_bounds__(FOLD_BLOCK) void fold_vecs_kernel(FoldVecs v, Fe<P> r) {
    fold_wave_prio();
    int k = 0;
#pragma unroll
    for (int j = 1; j < FOLD_VECS_MAX; j++)
        if (j < v.count && blockIdx.x >= v.first_block[j]) k = j;
    const uint4* __restrict__ a = v.a[k];
    const uint4* __restrict__ b = v.b[k];
    uint4* __restrict__ out = v.out[k];
    const size_t n = v.n[k];
    const size_t base = (size_t)(blockIdx.x - v.first_block[k]) * (FOLD_BLOCK * FOLD_VEC_E) + threadIdx.x;
    uint4 al[FOLD_VEC_E], ah[FOLD_VEC_E], bl[FOLD_VEC_E], bh[FOLD_VEC_E];
#pragma unroll
    for (int e = 0; e < FOLD_VEC_E; e++) {
        const size_t i = base + (size_t)e * FOLD_BLOCK;
        if (i < n) {
            al[e] = a[2 * i];
            ah[e] = a[2 * i + 1];
            bl[e] = b[2 * i];
            bh[e] = b[2 * i + 1];
        }
    }
#pragma unroll
    for (int e = 0; e < FOLD_VEC_E; e++) {
        const size_t i = base + (size_t)e * FOLD_BLOCK;
        if (i >= n) continue;
        Fe<P> x, y;
        x.l[0] = al[e].x; x.l[1] = al[e].y; x.l[2] = al[e].z; x.l[3] = al[e].w; x.l[4] = ah[e].x; x.l[5] = ah[e].y; x.l[6] = ah[e].z; x.l[7] = ah[e].w;
        y.l[0] = bl[e].x; y.l[1] = bl[e].y; y.l[2] = bl[e].z; y.l[3] = bl[e].w; y.l[4] = bh[e].x; y.l[5] = bh[e].y; y.l[6] = bh[e].z; y.l[7] = bh[e].w;
        x = fe_add<P>(x, fe_mul<P>(r, y));
        out[2 * i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
        out[2 * i + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    }
}

// grid of the one-lane-per-row kernels: a multiple of FOLD_XCDS workgroups, so that fold_row_block is a bijection onto [0, grid)
static unsigned fold_grid(size_t rows) {
    const unsigned nb = div_up(rows, FOLD_BLOCK);
    return (nb + FOLD_XCDS - 1) / FOLD_XCDS * FOLD_XCDS;
}

// ---- host side ---------------------------------------------------------------------------------------------
struct Key32 {
    uint64_t w[4];
    bool operator==(const Key32& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
};
struct Key32Hash {
    size_t operator()(const Key32& k) const {
        uint64_t h = k.w[0] * 0x9E3779B97F4A7C15ull;
        h ^= (k.w[1] + 0xBF58476D1CE4E5B9ull) * 0x94D049BB133111EBull;
        h ^= (k.w[2] + 0x2545F4914F6CDD1Dull) * 0xD6E8FEB86659FD93ull;
        h ^= (k.w[3] + 0x9E3779B97F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        return (size_t)(h ^ (h >> 29));
    }
};

template <class P>
static void dict_record(const uint64_t* mont256, uint32_t* rec) {
    Fe<P> v;
    for (int i = 0; i < 4; i++) { v.l[2 * i] = (uint32_t)mont256[i]; v.l[2 * i + 1] = (uint32_t)(mont256[i] >> 32); }
    for (int d = 0; d < 5; d++) v = fe_add<P>(v, v);  // c 2^256 -> c 2^261 mod p, canonical
    F29<P> f = f29_from_plain<P>(v.l);
    for (int i = 0; i < P29_STRIDE; i++) rec[i] = i < 9 ? f.l[i] : 0u;
}

static void upload_matrix(R1csShape& sh, int which, const uint64_t* indptr, const uint64_t* indices, const void* data, size_t ncols,
                          std::unordered_map<Key32, uint32_t, Key32Hash>& ids, std::vector<uint32_t>& dict_words) {
    const size_t rows = sh.num_cons;
    LURK_REQUIRE(indptr && indptr[0] == 0, "indptr must start at 0");
    const size_t nnz = indptr[rows];
    LURK_REQUIRE(nnz < ((size_t)1 << 32), "more than 2^32 - 1 non-zeros");
    LURK_REQUIRE(nnz == 0 || (indices && data), "null indices / data");
    std::vector<uint32_t> rp(rows + 1);
    for (size_t i = 0; i <= rows; i++) {
        LURK_REQUIRE(i == 0 || indptr[i] >= indptr[i - 1], "indptr must be non-decreasing");
        rp[i] = (uint32_t)indptr[i];
    }
    std::vector<uint2> ent(nnz);
    const uint64_t* d = (const uint64_t*)data;
    for (size_t k = 0; k < nnz; k++) {
        LURK_REQUIRE(indices[k] < ncols, "column index out of range");
        Key32 key{{d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]}};
        auto it = ids.find(key);
        if (it == ids.end()) {
            uint32_t id = (uint32_t)ids.size();
            it = ids.emplace(key, id).first;
            dict_words.resize(dict_words.size() + P29_STRIDE);
            uint32_t* rec = dict_words.data() + (size_t)id * P29_STRIDE;
            if (sh.field_id == 0) dict_record<PallasFp>(key.w, rec);
            else if (sh.field_id == 1) dict_record<PallasFq>(key.w, rec);
            else dict_record<Bn254Fr>(key.w, rec);
        }
        ent[k] = make_uint2((uint32_t)indices[k], it->second);
    }
    CsrDev& m = sh.m[which];
    m.nnz = nnz;
    m.rowptr.alloc(rp.size() * 4);
    m.ent.alloc(nnz * 8);
    LURK_HIP_CHECK(hipMemcpy(m.rowptr.p, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    if (nnz) LURK_HIP_CHECK(hipMemcpy(m.ent.p, ent.data(), nnz * 8, hipMemcpyHostToDevice));
}

static R1csDev dev_view(const R1csShape& sh) {
    R1csDev d;
    d.a = CsrView{sh.m[0].rowptr.as<uint32_t>(), sh.m[0].ent.as<uint2>()};
    d.b = CsrView{sh.m[1].rowptr.as<uint32_t>(), sh.m[1].ent.as<uint2>()};
    d.c = CsrView{sh.m[2].rowptr.as<uint32_t>(), sh.m[2].ent.as<uint2>()};
    d.dict = sh.dict.as<uint32_t>();
    d.dict_size = sh.dict_size;
    d.rows = sh.num_cons;
    d.long_rows = sh.long_rows.as<uint32_t>();
    d.n_long = (uint32_t)sh.n_long;
    return d;
}

template <class P>
static void multiply_vec(const R1csShape& sh, const void* d_z, void* az, void* bz, void* cz, hipStream_t s) {
    if (!sh.num_cons) return;
    ProfScope ps("r1cs_multiply_vec", s);
    const R1csDev d = dev_view(sh);
    if (sh.n_long)  // first: its few latency-bound waves then run beside the lane-per-row launch that follows
        hipLaunchKernelGGL((r1cs_multiply_vec_kernel<P, true>), dim3(div_up(sh.n_long * FOLD_GROUP, FOLD_BLOCK)), dim3(FOLD_BLOCK), 0, s, d,
                           (const Fe<P>*)d_z, (Fe<P>*)az, (Fe<P>*)bz, (Fe<P>*)cz);
    hipLaunchKernelGGL((r1cs_multiply_vec_kernel<P, false>), dim3(fold_grid(sh.num_cons)), dim3(FOLD_BLOCK), 0, s, d, (const Fe<P>*)d_z,
                       (Fe<P>*)az, (Fe<P>*)bz, (Fe<P>*)cz);
    LURK_HIP_CHECK(hipGetLastError());
}
template <class P>
static void cross_term(const R1csShape& sh, const void* d_z1, const void* d_z2, void* d_t, hipStream_t s) {
    if (!sh.num_cons) return;
    ProfScope ps("r1cs_cross_term", s);
    const R1csDev d = dev_view(sh);
    const unsigned long_blocks = sh.n_long ? div_up(sh.n_long * FOLD_GROUP, FOLD_BLOCK) : 0;
    hipLaunchKernelGGL((r1cs_cross_term_kernel<P>), dim3(long_blocks + fold_grid(sh.num_cons)), dim3(FOLD_BLOCK), 0, s, d, (const Fe<P>*)d_z1,
                       (const Fe<P>*)d_z2, sh.num_vars, (Fe<P>*)d_t, long_blocks);
    LURK_HIP_CHECK(hipGetLastError());
}
template <class P>
static void cross_term_cached(const R1csShape& sh, const void* d_z2, void* az1, void* bz1, void* cz1, const void* u1_32, const void* prev_a, const void* prev_b,
                              const void* prev_c, const void* r_prev32, void* d_t, void* az2, void* bz2, void* cz2, hipStream_t s) {
    if (!sh.num_cons) return;
    ProfScope ps("r1cs_cross_term", s);
    const R1csDev d = dev_view(sh);
    const unsigned long_blocks = sh.n_long ? div_up(sh.n_long * FOLD_GROUP, FOLD_BLOCK) : 0;
    CachedProducts<P> cp;
    cp.a1 = (Fe<P>*)az1;
    cp.b1 = (Fe<P>*)bz1;
    cp.c1 = (Fe<P>*)cz1;
    cp.prev_a = (const Fe<P>*)prev_a;
    cp.prev_b = (const Fe<P>*)prev_b;
    cp.prev_c = (const Fe<P>*)prev_c;
    cp.r_prev = fe_zero<P>();
    if (r_prev32) memcpy(cp.r_prev.l, r_prev32, 32);
    memcpy(cp.u1.l, u1_32, 32);
    hipLaunchKernelGGL((r1cs_cross_term_cached_kernel<P>), dim3(long_blocks + fold_grid(sh.num_cons)), dim3(FOLD_BLOCK), 0, s, d, (const Fe<P>*)d_z2, cp,
                       sh.num_vars, (Fe<P>*)d_t, (Fe<P>*)az2, (Fe<P>*)bz2, (Fe<P>*)cz2, long_blocks);
    LURK_HIP_CHECK(hipGetLastError());
}
template <class P>
static void fold_vecs(int count, const void* const* a, const void* const* b, const size_t* n, void* const* out, const void* r32, hipStream_t s) {
    FoldVecs v;
    memset(&v, 0, sizeof(v));
    unsigned blocks = 0;
    for (int k = 0; k < count; k++) {
        if (!n[k]) continue;
        v.a[v.count] = (const uint4*)a[k];
        v.b[v.count] = (const uint4*)b[k];
        v.out[v.count] = (uint4*)out[k];
        v.n[v.count] = n[k];
        v.first_block[v.count] = blocks;
        blocks += div_up(n[k], (size_t)FOLD_BLOCK * FOLD_VEC_E);
        v.count++;
    }
    if (!v.count) return;
    v.first_block[v.count] = blocks;
    Fe<P> r;
    memcpy(r.l, r32, 32);
    ProfScope ps("fold_vec", s);
    hipLaunchKernelGGL((fold_vecs_kernel<P>), dim3(blocks), dim3(FOLD_BLOCK), 0, s, v, r);
    LURK_HIP_CHECK(hipGetLastError());
}
template <class P>
static void fold_vec(const void* a, const void* b, const void* r32, size_t n, void* out, hipStream_t s) {
    if (!n) return;
    Fe<P> r;
    memcpy(r.l, r32, 32);
    ProfScope ps("fold_vec", s);
    const unsigned blocks = div_up(n, (size_t)FOLD_BLOCK * FOLD_VEC_E);
    hipLaunchKernelGGL((fold_vec_kernel<P>), dim3(blocks), dim3(FOLD_BLOCK), 0, s, (const uint4*)a, (const uint4*)b, r, n, (uint4*)out);
    LURK_HIP_CHECK(hipGetLastError());
}

}  // namespace lurk

using namespace lurk;

struct lurk_hip_r1cs {
    R1csShape sh;  // immutable after creation: calls on one shape from any thread / stream are independent
};

extern "C" {

int lurk_hip_r1cs_create(lurk_hip_r1cs** out, int field_id, size_t num_cons, size_t num_vars, size_t num_io, const uint64_t* a_indptr,
                         const uint64_t* a_indices, const void* a_data, const uint64_t* b_indptr, const uint64_t* b_indices, const void* b_data,
                         const uint64_t* c_indptr, const uint64_t* c_indices, const void* c_data) {
    return guarded([&] {
        LURK_REQUIRE(out, "null output handle");
        *out = nullptr;
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(num_vars + 1 + num_io < ((size_t)1 << 32), "z has more than 2^32 - 1 entries");
        auto h = std::make_unique<lurk_hip_r1cs>();
        R1csShape& sh = h->sh;
        sh.field_id = field_id;
        sh.num_cons = num_cons;
        sh.num_vars = num_vars;
        sh.num_io = num_io;
        LURK_HIP_CHECK(hipGetDevice(&sh.device));
        std::unordered_map<Key32, uint32_t, Key32Hash> ids;
        std::vector<uint32_t> dict_words;
        const size_t ncols = num_vars + 1 + num_io;
        upload_matrix(sh, 0, a_indptr, a_indices, a_data, ncols, ids, dict_words);
        upload_matrix(sh, 1, b_indptr, b_indices, b_data, ncols, ids, dict_words);
        upload_matrix(sh, 2, c_indptr, c_indices, c_data, ncols, ids, dict_words);
        sh.dict_size = ids.size();
        {   // closing record: the Montgomery one (1 * 2^256 mod p as the ABI stores it)
            dict_words.resize(dict_words.size() + P29_STRIDE);
            uint32_t* rec = dict_words.data() + sh.dict_size * P29_STRIDE;
            uint64_t one[4];
            if (field_id == 0) { Fe<PallasFp> o = fe_one<PallasFp>(); memcpy(one, o.l, 32); dict_record<PallasFp>(one, rec); }
            else if (field_id == 1) { Fe<PallasFq> o = fe_one<PallasFq>(); memcpy(one, o.l, 32); dict_record<PallasFq>(one, rec); }
            else { Fe<Bn254Fr> o = fe_one<Bn254Fr>(); memcpy(one, o.l, 32); dict_record<Bn254Fr>(one, rec); }
        }
        {   // rows the lane-per-row kernels leave to the wave-per-row ones
            std::vector<uint32_t> lr;
            for (size_t i = 0; i < num_cons; i++)
                if (a_indptr[i + 1] - a_indptr[i] > FOLD_LONG || b_indptr[i + 1] - b_indptr[i] > FOLD_LONG || c_indptr[i + 1] - c_indptr[i] > FOLD_LONG)
                    lr.push_back((uint32_t)i);
            sh.n_long = lr.size();
            sh.long_rows.alloc(lr.size() * 4);
            if (!lr.empty()) LURK_HIP_CHECK(hipMemcpy(sh.long_rows.p, lr.data(), lr.size() * 4, hipMemcpyHostToDevice));
        }
        sh.dict.alloc(dict_words.size() * 4);
        if (!dict_words.empty()) LURK_HIP_CHECK(hipMemcpy(sh.dict.p, dict_words.data(), dict_words.size() * 4, hipMemcpyHostToDevice));
        *out = h.release();
    });
}

int lurk_hip_r1cs_destroy(lurk_hip_r1cs* shape) {
    if (!shape) return 0;
    return guarded([&] {
        DeviceGuard dg(shape->sh.device);
        delete shape;
    });
}

int lurk_hip_r1cs_dims(const lurk_hip_r1cs* shape, int* field_id, size_t* num_cons, size_t* num_vars, size_t* num_io) {
    return guarded([&] {
        LURK_REQUIRE(shape, "null shape");
        if (field_id) *field_id = shape->sh.field_id;
        if (num_cons) *num_cons = shape->sh.num_cons;
        if (num_vars) *num_vars = shape->sh.num_vars;
        if (num_io) *num_io = shape->sh.num_io;
    });
}

int lurk_hip_r1cs_device(const lurk_hip_r1cs* shape, int* device) {
    return guarded([&] {
        LURK_REQUIRE(shape && device, "null argument");
        *device = shape->sh.device;
    });
}

int lurk_hip_r1cs_info(const lurk_hip_r1cs* shape, size_t* nnz_a, size_t* nnz_b, size_t* nnz_c, size_t* distinct_coefficients) {
    return guarded([&] {
        LURK_REQUIRE(shape, "null shape");
        if (nnz_a) *nnz_a = shape->sh.m[0].nnz;
        if (nnz_b) *nnz_b = shape->sh.m[1].nnz;
        if (nnz_c) *nnz_c = shape->sh.m[2].nnz;
        if (distinct_coefficients) *distinct_coefficients = shape->sh.dict_size;
    });
}

int lurk_hip_r1cs_multiply_vec_dev(const lurk_hip_r1cs* shape, const void* d_z, void* d_az, void* d_bz, void* d_cz, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(shape && d_z && d_az && d_bz && d_cz, "null argument");
        DeviceGuard dg(shape->sh.device);
        const R1csShape& sh = shape->sh;
        if (sh.field_id == 0) multiply_vec<PallasFp>(sh, d_z, d_az, d_bz, d_cz, (hipStream_t)stream);
        else if (sh.field_id == 1) multiply_vec<PallasFq>(sh, d_z, d_az, d_bz, d_cz, (hipStream_t)stream);
        else multiply_vec<Bn254Fr>(sh, d_z, d_az, d_bz, d_cz, (hipStream_t)stream);
    });
}

int lurk_hip_r1cs_cross_term_dev(lurk_hip_r1cs* shape, const void* d_z1, const void* d_z2, void* d_t, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(shape && d_z1 && d_z2 && d_t, "null argument");
        DeviceGuard dg(shape->sh.device);
        const R1csShape& sh = shape->sh;
        if (sh.field_id == 0) cross_term<PallasFp>(sh, d_z1, d_z2, d_t, (hipStream_t)stream);
        else if (sh.field_id == 1) cross_term<PallasFq>(sh, d_z1, d_z2, d_t, (hipStream_t)stream);
        else cross_term<Bn254Fr>(sh, d_z1, d_z2, d_t, (hipStream_t)stream);
    });
}

int lurk_hip_r1cs_cross_term_cached_dev(lurk_hip_r1cs* shape, const void* d_z2, void* d_az1, void* d_bz1, void* d_cz1, const void* u1_32_mont,
                                        const void* d_az2_prev, const void* d_bz2_prev, const void* d_cz2_prev, const void* r_prev32_mont, void* d_t,
                                        void* d_az2, void* d_bz2, void* d_cz2, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(shape && d_z2 && d_az1 && d_bz1 && d_cz1 && u1_32_mont && d_t && d_az2 && d_bz2 && d_cz2, "null argument");
        const bool prev = d_az2_prev || d_bz2_prev || d_cz2_prev || r_prev32_mont;
        LURK_REQUIRE(!prev || (d_az2_prev && d_bz2_prev && d_cz2_prev && r_prev32_mont), "the previous step's products and its challenge go together");
        LURK_REQUIRE(!prev || (d_az2_prev != d_az2 && d_bz2_prev != d_bz2 && d_cz2_prev != d_cz2), "the previous products are read while the new ones are written: two buffers");
        DeviceGuard dg(shape->sh.device);
        const R1csShape& sh = shape->sh;
        if (sh.field_id == 0)
            cross_term_cached<PallasFp>(sh, d_z2, d_az1, d_bz1, d_cz1, u1_32_mont, d_az2_prev, d_bz2_prev, d_cz2_prev, r_prev32_mont, d_t, d_az2, d_bz2, d_cz2, (hipStream_t)stream);
        else if (sh.field_id == 1)
            cross_term_cached<PallasFq>(sh, d_z2, d_az1, d_bz1, d_cz1, u1_32_mont, d_az2_prev, d_bz2_prev, d_cz2_prev, r_prev32_mont, d_t, d_az2, d_bz2, d_cz2, (hipStream_t)stream);
        else
            cross_term_cached<Bn254Fr>(sh, d_z2, d_az1, d_bz1, d_cz1, u1_32_mont, d_az2_prev, d_bz2_prev, d_cz2_prev, r_prev32_mont, d_t, d_az2, d_bz2, d_cz2, (hipStream_t)stream);
    });
}

int lurk_hip_fold_vecs_dev(int field_id, int count, const void* const* d_a, const void* const* d_b, const size_t* n, void* const* d_out,
                           const void* r32_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(count >= 0 && count <= FOLD_VECS_MAX, "at most 8 vectors per launch");
        LURK_REQUIRE(count == 0 || (d_a && d_b && n && d_out), "null argument");
        LURK_REQUIRE(r32_mont, "null challenge");
        size_t blocks = 0;
        for (int k = 0; k < count; k++) {
            LURK_REQUIRE(n[k] == 0 || (d_a[k] && d_b[k] && d_out[k]), "null buffer");
            blocks += div_up(n[k], (size_t)FOLD_BLOCK * FOLD_VEC_E);
        }
        LURK_REQUIRE(blocks < ((size_t)1 << 31), "too many elements for one launch");
        if (field_id == 0) fold_vecs<PallasFp>(count, d_a, d_b, n, d_out, r32_mont, (hipStream_t)stream);
        else if (field_id == 1) fold_vecs<PallasFq>(count, d_a, d_b, n, d_out, r32_mont, (hipStream_t)stream);
        else fold_vecs<Bn254Fr>(count, d_a, d_b, n, d_out, r32_mont, (hipStream_t)stream);
    });
}

// host-pointer forms (small inputs, tests, the C++ mirror): copy in, run, copy out
int lurk_hip_r1cs_multiply_vec(const lurk_hip_r1cs* shape, const void* z, void* az, void* bz, void* cz) {
    return guarded([&] {
        LURK_REQUIRE(shape && z && az && bz && cz, "null argument");
        DeviceGuard dg(shape->sh.device);
        const R1csShape& sh = shape->sh;
        const size_t ncols = sh.num_vars + 1 + sh.num_io, m = sh.num_cons;
        DevBuf d_z(ncols * 32), d_a(m * 32), d_b(m * 32), d_c(m * 32);
        LURK_HIP_CHECK(hipMemcpy(d_z.p, z, ncols * 32, hipMemcpyHostToDevice));
        LURK_REQUIRE(lurk_hip_r1cs_multiply_vec_dev(shape, d_z.p, d_a.p, d_b.p, d_c.p, nullptr) == 0, lurk_hip_last_error());
        LURK_HIP_CHECK(hipDeviceSynchronize());
        LURK_HIP_CHECK(hipMemcpy(az, d_a.p, m * 32, hipMemcpyDeviceToHost));
        LURK_HIP_CHECK(hipMemcpy(bz, d_b.p, m * 32, hipMemcpyDeviceToHost));
        LURK_HIP_CHECK(hipMemcpy(cz, d_c.p, m * 32, hipMemcpyDeviceToHost));
    });
}

int lurk_hip_r1cs_cross_term(lurk_hip_r1cs* shape, const void* z1, const void* z2, void* t) {
    return guarded([&] {
        LURK_REQUIRE(shape && z1 && z2 && t, "null argument");
        DeviceGuard dg(shape->sh.device);
        const R1csShape& sh = shape->sh;
        const size_t ncols = sh.num_vars + 1 + sh.num_io, m = sh.num_cons;
        DevBuf d_z1(ncols * 32), d_z2(ncols * 32), d_t(m * 32);
        LURK_HIP_CHECK(hipMemcpy(d_z1.p, z1, ncols * 32, hipMemcpyHostToDevice));
        LURK_HIP_CHECK(hipMemcpy(d_z2.p, z2, ncols * 32, hipMemcpyHostToDevice));
        LURK_REQUIRE(lurk_hip_r1cs_cross_term_dev(shape, d_z1.p, d_z2.p, d_t.p, nullptr) == 0, lurk_hip_last_error());
        LURK_HIP_CHECK(hipDeviceSynchronize());
        LURK_HIP_CHECK(hipMemcpy(t, d_t.p, m * 32, hipMemcpyDeviceToHost));
    });
}

int lurk_hip_fold_vec(int field_id, const void* a, const void* b, const void* r32_mont, size_t n, void* out) {
    return guarded([&] {
        LURK_REQUIRE(n == 0 || (a && b && out), "null buffer");
        DevBuf d_a(n * 32), d_b(n * 32), d_o(n * 32);
        if (n) {
            LURK_HIP_CHECK(hipMemcpy(d_a.p, a, n * 32, hipMemcpyHostToDevice));
            LURK_HIP_CHECK(hipMemcpy(d_b.p, b, n * 32, hipMemcpyHostToDevice));
        }
        LURK_REQUIRE(lurk_hip_fold_vec_dev(field_id, d_a.p, d_b.p, r32_mont, n, d_o.p, nullptr) == 0, lurk_hip_last_error());
        LURK_HIP_CHECK(hipDeviceSynchronize());
        if (n) LURK_HIP_CHECK(hipMemcpy(out, d_o.p, n * 32, hipMemcpyDeviceToHost));
    });
}

int lurk_hip_fold_vec_dev(int field_id, const void* d_a, const void* d_b, const void* r32_mont, size_t n, void* d_out, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(n == 0 || (d_a && d_b && d_out), "null buffer");
        LURK_REQUIRE(r32_mont, "null challenge");
        if (field_id == 0) fold_vec<PallasFp>(d_a, d_b, r32_mont, n, d_out, (hipStream_t)stream);
        else if (field_id == 1) fold_vec<PallasFq>(d_a, d_b, r32_mont, n, d_out, (hipStream_t)stream);
        else fold_vec<Bn254Fr>(d_a, d_b, r32_mont, n, d_out, (hipStream_t)stream);
    });
}

}  // extern "C"

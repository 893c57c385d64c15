// sumcheck.hip - the data-parallel half of Spartan's sum-check, for CompressedSNARK::prove.
//
// Reference call sites: CompressedSNARK::prove as lurk-beta reaches it (/root/reference/src/proof/nova.rs:341-356,
// /root/reference/src/proof/supernova.rs:293-302) -> arecibo spartan::snark::RelaxedR1CSSNARK::prove -> SumcheckProof::
// prove_cubic_with_additive_term (outer: eq(tau) * (Az * Bz - (u Cz + E))) and prove_quad (inner: poly_ABC * z), arecibo being the
// un-vendored `nova` dependency (/root/reference/Cargo.toml:128): the published Spartan prover, restated in oracle/pyref.py
// (parity unpinned: no proof bytes exist upstream).  Per round the prover needs
//     e0 = sum_i comb(P[i]),  e2 = sum_i comb(2 P[h+i] - P[i]),  e3 = sum_i comb(3 P[h+i] - 2 P[i])        (h = len / 2)
// then, with the verifier's challenge r, binds the top variable of every table: P[i] <- P[i] + r (P[h+i] - P[i]).
// The transcript (Keccak) stays on the host; the tables (2^20 .. 2^24 field elements each) stay in HBM.  One kernel per round
// does BOTH the bind with the previous challenge and the evaluation sums of the next round, so a round reads each table once
// (4 x 32 B per element) and writes half of it back: HBM-bound, 160 B per element-row of the cubic round.
#include <memory>

#include "common.hpp"
#include "field.cuh"
#include "sumcheck_host.hpp"

namespace lurk {

constexpr int SC_BLOCK = 256;


// workgroup tree sum of NV values per thread; thread 0 writes the block's partial sums.  With a ticket counter (final != nullptr) the
// LAST workgroup to arrive (agent-scope ticket: release before, acquire after - cdna_hip_programming.md guideline 16) sums every
// workgroup's partials and stores the NV totals straight into `final` - pinned host memory - so that a round hands the host 96 bytes
// behind one synchronisation instead of a pageable copy of up to 2 048 x NV partial sums and their summation on the host (0.09 ms per
// round of a 2^20-row proof, 85 rounds).
template <class F, int NV>
__device__ __forceinline__ void sc_block_sum(Fe<F>* v, Fe<F>* __restrict__ partial, uint32_t* __restrict__ counter, Fe<F>* __restrict__ final) {
    __shared__ uint4 raw[SC_BLOCK * NV * 2];
    __shared__ uint32_t sh_ticket;
    Fe<F>* sh = reinterpret_cast<Fe<F>*>(raw);
    const int t = threadIdx.x;
    auto tree = [&] {
#pragma unroll
        for (int k = 0; k < NV; k++) sh[k * SC_BLOCK + t] = v[k];
        __syncthreads();
        for (int s = SC_BLOCK / 2; s >= 1; s >>= 1) {
            if (t < s) {
#pragma unroll
                for (int k = 0; k < NV; k++) sh[k * SC_BLOCK + t] = fe_add<F>(sh[k * SC_BLOCK + t], sh[k * SC_BLOCK + t + s]);
            }
            __syncthreads();
        }
    };
    tree();
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) partial[(size_t)blockIdx.x * NV + k] = sh[k * SC_BLOCK];
        if (final) {
            __threadfence();  // the partial sums are visible device-wide before the ticket is taken
            sh_ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!final) return;
    __syncthreads();
    if (sh_ticket != gridDim.x - 1) return;
    __threadfence();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = fe_zero<F>();
    for (unsigned b = t; b < gridDim.x; b += SC_BLOCK) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = fe_add<F>(v[k], partial[(size_t)b * NV + k]);
    }
    __syncthreads();  // (everybody is past its reads of sh from the first tree)
    tree();
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) final[k] = sh[k * SC_BLOCK];
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next round
        __threadfence_system();
    }
}

// NP tables of `len` elements (len = 2 m).  BIND: first P[i] <- bind(P[i], P[m+i], r) for i < m (in place), then the
// evaluation sums over the bound tables (length m, halves of q = m / 2); !BIND: sums over the tables as they are (h = m).
// Cubic (NP = 4): e0, e2, e3 with comb = a (b c - d); quadratic (NP = 2): e0, e2 with comb = a b.
template <class F, int NP, bool BIND>
__global__ __launch_bounds__(SC_BLOCK) void sumcheck_round_kernel(Fe<F>* p0, Fe<F>* p1, Fe<F>* p2, Fe<F>* p3, size_t len, Fe<F> r,
                                                                    Fe<F>* __restrict__ partial, uint32_t* __restrict__ counter, Fe<F>* __restrict__ final) {
    constexpr int NV = NP == 4 ? 3 : 2;
    Fe<F>* P[4] = {p0, p1, p2, p3};
    Fe<F> acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = fe_zero<F>();
    const size_t m = len / 2, h = BIND ? m / 2 : m;  // pairs (i, i + h) of the tables the sums run over
    for (size_t i = (size_t)blockIdx.x * SC_BLOCK + threadIdx.x; i < (h ? h : 1); i += (size_t)gridDim.x * SC_BLOCK) {
        Fe<F> lo[NP], hi[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) {
            if (BIND) {
                if (h == 0) {  // len == 2: the last bind, nothing left to sum
                    P[k][0] = sc_bind<F>(P[k][0], P[k][1], r);
                    continue;
                }
                lo[k] = sc_bind<F>(P[k][i], P[k][m + i], r);
                hi[k] = sc_bind<F>(P[k][h + i], P[k][m + h + i], r);
                P[k][i] = lo[k];
                P[k][h + i] = hi[k];
            } else {
                lo[k] = P[k][i];
                hi[k] = P[k][h + i];
            }
        }
        if (BIND && h == 0) break;
        Fe<F> b2[NP], b3[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const Fe<F> d = fe_sub<F>(hi[k], lo[k]);
            b2[k] = fe_add<F>(hi[k], d);  // 2 hi - lo
            if (NP == 4) b3[k] = fe_add<F>(b2[k], d);  // 3 hi - 2 lo
        }
        if (NP == 4) {
            acc[0] = fe_add<F>(acc[0], sc_comb_cubic<F>(lo[0], lo[1], lo[2], lo[3]));
            acc[1] = fe_add<F>(acc[1], sc_comb_cubic<F>(b2[0], b2[1], b2[2], b2[3]));
            acc[2] = fe_add<F>(acc[2], sc_comb_cubic<F>(b3[0], b3[1], b3[2], b3[3]));
        } else {
            acc[0] = fe_add<F>(acc[0], fe_mul<F>(lo[0], lo[1]));
            acc[1] = fe_add<F>(acc[1], fe_mul<F>(b2[0], b2[1]));
        }
    }
    sc_block_sum<F, NV>(acc, partial, counter, final);
}

// EqPolynomial::evals: out[b] = prod_j (b_j ? r_j : 1 - r_j), r_0 the most significant bit of b.
// The product splits at bit EQ_LO_BITS: out[b] = hi(b >> lo) * low[b & (2^lo - 1)].  A workgroup builds the 2^lo-entry table of the low
// factors in LDS by doubling (one product per entry: t * r and t - t * r) and walks chunks of 2^lo consecutive outputs; a lane pays the
// ell - lo products of its chunk's high factor once and then one product per output: ~5 products per output at ell = 20 where the
// direct form takes 20 (0.17 -> 0.05 ms for 2^20 entries; the three sum-checks of a proof ask for eight of these tables).
constexpr int EQ_LO_BITS = 10;
template <class F>
__global__ __launch_bounds__(SC_BLOCK) void eq_evals_kernel(const Fe<F>* __restrict__ r, int ell, Fe<F>* __restrict__ out) {
    extern __shared__ uint4 raw[];
    Fe<F>* low = reinterpret_cast<Fe<F>*>(raw);  // [2^lo]
    const int lo = ell < EQ_LO_BITS ? ell : EQ_LO_BITS, hi = ell - lo;
    const uint32_t L = 1u << lo;
    if (threadIdx.x == 0) low[0] = fe_one<F>();
    __syncthreads();
    for (int t = 0; t < lo; t++) {  // after step t the entries at stride 2^(lo - 1 - t) hold the products over r_hi .. r_(hi + t)
        const uint32_t stride = L >> t, half = stride >> 1;
        const Fe<F> rt = r[hi + t];
        for (uint32_t i = threadIdx.x; i < (1u << t); i += SC_BLOCK) {
            const Fe<F> v = low[i * stride], vr = fe_mul<F>(v, rt);
            low[i * stride + half] = vr;
            low[i * stride] = fe_sub<F>(v, vr);
        }
        __syncthreads();
    }
    const size_t chunks = (size_t)1 << hi;
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        Fe<F> h = fe_one<F>();
        for (int j = 0; j < hi; j++) {
            const Fe<F> rj = r[j];
            h = fe_mul<F>(h, ((c >> (hi - 1 - j)) & 1) ? rj : fe_sub<F>(fe_one<F>(), rj));
        }
        for (uint32_t k = threadIdx.x; k < L; k += SC_BLOCK) out[c * L + k] = fe_mul<F>(h, low[k]);
    }
}

// What the rounds of one sum-check share: the workgroups' partial sums, the ticket counter of the last-workgroup reduction and the
// pinned host words the totals land in.
struct SumcheckScratch {
    void* partial = nullptr;
    uint32_t* counter = nullptr;
    void* host_final = nullptr;  // 3 x 32 B, pinned (device-visible)
    hipStream_t s = nullptr;
    explicit SumcheckScratch(hipStream_t s_) : s(s_) {
        stream_pool_retain();
        const size_t cap = (size_t)num_cus() * 8;
        LURK_HIP_CHECK(hipMallocAsync(&partial, cap * 3 * 32 + 256, s));
        counter = (uint32_t*)((char*)partial + cap * 3 * 32);
        LURK_HIP_CHECK(hipMemsetAsync(counter, 0, 4, s));
        if (hipHostMalloc(&host_final, 96, hipHostMallocDefault) != hipSuccess) {
            (void)hipFreeAsync(partial, s);
            throw HipFailure{LURK_HIP_ERR_HIP, "hipHostMalloc of the round totals failed"};
        }
    }
    ~SumcheckScratch() {
        (void)hipStreamSynchronize(s);  // (a kernel may still hold the pinned words)
        (void)hipFreeAsync(partial, s);
        (void)hipHostFree(host_final);
    }
    SumcheckScratch(const SumcheckScratch&) = delete;
};

struct SumcheckCachedScratch {
    std::mutex mu;  // rounds that share a stream are ordered on it; their host sides take turns here
    SumcheckScratch sc;
    explicit SumcheckCachedScratch(hipStream_t s) : sc(s) {}
};
static SumcheckScratch* sumcheck_cached_scratch(hipStream_t s, std::unique_lock<std::mutex>& lk) {
    static std::mutex map_mu;
    static auto& cache = *new std::map<std::pair<int, hipStream_t>, SumcheckCachedScratch*>();  // never torn down (process lifetime)
    SumcheckCachedScratch* e;
    {
        std::lock_guard<std::mutex> g(map_mu);
        auto key = std::make_pair(current_device(), s);
        auto it = cache.find(key);
        if (it == cache.end()) it = cache.emplace(key, new SumcheckCachedScratch(s)).first;
        e = it->second;
    }
    lk = std::unique_lock<std::mutex>(e->mu);
    return &e->sc;
}

template <class F>
static void sumcheck_round(int np, void* const* d_polys, size_t len, const void* r32_mont, void* evals_out, hipStream_t s, SumcheckScratch* sc = nullptr) {
    LURK_REQUIRE(len >= 2 && (len & (len - 1)) == 0, "table length must be a power of two >= 2");
    const bool bind = r32_mont != nullptr;
    const size_t work = bind ? len / 4 : len / 2;
    unsigned blocks = work ? div_up(work, SC_BLOCK) : 1;
    const unsigned cap = (unsigned)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    const int nv = np == 4 ? 3 : 2;
    Fe<F> r = fe_zero<F>();
    if (bind) memcpy(r.l, r32_mont, 32);
    // a caller that drives its own round loop through the per-round entry point gets ONE scratch per (device, stream), made at its
    // first round and kept: no pinned allocation, no synchronisation beyond the one that hands the round's totals to the host - and
    // none at all for the last bind (evals_out == NULL)
    std::unique_lock<std::mutex> cached_lk;
    if (!sc) sc = sumcheck_cached_scratch(s, cached_lk);
    Fe<F>* partial = (Fe<F>*)sc->partial;
    uint32_t* counter = evals_out ? sc->counter : nullptr;
    Fe<F>* fin = evals_out ? (Fe<F>*)sc->host_final : nullptr;
    Fe<F>* p[4] = {(Fe<F>*)d_polys[0], (Fe<F>*)d_polys[1], np == 4 ? (Fe<F>*)d_polys[2] : nullptr, np == 4 ? (Fe<F>*)d_polys[3] : nullptr};
    {
        ProfScope ps("sumcheck_round", s);
        if (np == 4 && bind) hipLaunchKernelGGL((sumcheck_round_kernel<F, 4, true>), dim3(blocks), dim3(SC_BLOCK), 0, s, p[0], p[1], p[2], p[3], len, r, partial, counter, fin);
        else if (np == 4) hipLaunchKernelGGL((sumcheck_round_kernel<F, 4, false>), dim3(blocks), dim3(SC_BLOCK), 0, s, p[0], p[1], p[2], p[3], len, r, partial, counter, fin);
        else if (bind) hipLaunchKernelGGL((sumcheck_round_kernel<F, 2, true>), dim3(blocks), dim3(SC_BLOCK), 0, s, p[0], p[1], p[2], p[3], len, r, partial, counter, fin);
        else hipLaunchKernelGGL((sumcheck_round_kernel<F, 2, false>), dim3(blocks), dim3(SC_BLOCK), 0, s, p[0], p[1], p[2], p[3], len, r, partial, counter, fin);
    }
    LURK_HIP_CHECK(hipGetLastError());
    if (evals_out) {  // the round polynomial goes into the host transcript: this is the round's one synchronisation
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(evals_out, sc->host_final, (size_t)nv * 32);
    }
}


// ---- the last rounds on the host -------------------------------------------------------------------------------------------------
// A round over short tables is all latency on the device: one launch, ~14 dependent 8 x 32 products on a lane (21 us), the last
// workgroup's sums, one synchronisation - ~45 us per round whatever the length, and the three sum-checks of a proof run 85 rounds.  A
// host core does a field product in ~45 ns: from SC_HOST_TAIL elements per table on, the tables come to the host once (one copy per
// table, one synchronisation) and the remaining rounds - the same bind, the same evaluation sums, in the same exact arithmetic - cost
// 70, 35, 17, ... us.  The device tables are left as they were at that point (sumcheck_prove consumes them).
static size_t sumcheck_host_tail_len() {
    const char* e = getenv("LURK_SUMCHECK_HOST_TAIL_LOG");  // log2 of the table length from which the rounds run on the host; 0 = never; read per call
    const int v = e ? atoi(e) : 8;
    return v <= 0 ? 0 : (size_t)1 << (v > 16 ? 16 : v);
}
// ---- a whole sum-check as host code of the library (arecibo SumcheckProof::prove_quad / prove_cubic_with_additive_term behind
// /root/reference/src/proof/nova.rs:341-356): the round loop of lurk_beta_amd/sumcheck.py: prove - one launch per round, the round
// polynomial interpolated from its evaluations at 0, 2 (, 3) and the running claim, the transcript's challenge from a callback.
// Batched form (ninst > 1): sum_i coeff_i * sum_x comb(tables of instance i) with ONE challenge per round shared by every instance - the
// evaluation-claim batching of arecibo's snark.rs and the outer / inner sum-checks of its BatchedRelaxedR1CSSNARK
// (/root/reference/src/proof/supernova.rs:110, 293-302): per round one launch per instance, the evaluations combined with the
// coefficients on the host, one interpolation, one callback.  d_polys holds ninst x np tables, instance-major, all of length n.
template <class F>
static void sumcheck_prove(int np, size_t ninst, void* const* d_polys, size_t n, const void* coeffs32_canonical, const void* claim32_canonical,
                           lurk_hip_sumcheck_challenge_fn challenge, void* user, uint64_t* out_polys, uint64_t* out_finals, void* out_claim32, hipStream_t s) {
    const bool cubic = np == 4;
    std::vector<Fe<F>> coeff(ninst, fe_one<F>());
    for (size_t i = 0; coeffs32_canonical && i < ninst; i++) {
        Fe<F> c;
        memcpy(c.l, (const char*)coeffs32_canonical + 32 * i, 32);
        LURK_REQUIRE(!fe_canonical_ge_mod<F>(c.l), "a batching coefficient is not reduced modulo the field order");
        coeff[i] = fe_to_mont<F>(c);
    }
    const int nv = cubic ? 3 : 2, ncoef = cubic ? 4 : 3;
    const Fe<F> two = fe_from_u64<F>(2), three = fe_from_u64<F>(3), inv2 = fe_inv<F>(two), inv6 = fe_inv<F>(fe_from_u64<F>(6));
    Fe<F> claim;
    memcpy(claim.l, claim32_canonical, 32);
    LURK_REQUIRE(!fe_canonical_ge_mod<F>(claim.l), "the claim is not reduced modulo the field order");
    claim = fe_to_mont<F>(claim);
    size_t length = n;
    Fe<F> r = fe_zero<F>();
    bool have_r = false;
    int j = 0;
    // the (device, stream)'s scratch, made once and kept: a proof runs three of these loops
    std::unique_lock<std::mutex> scratch_lk;
    SumcheckScratch& scratch = *sumcheck_cached_scratch(s, scratch_lk);
    // the scratch is cached per (device, stream): the last-workgroup ticket counter is reset by the workgroup that takes the last ticket,
    // so a round that faulted midway would leave it non-zero for every later sum-check on this stream - it starts every call at zero
    LURK_HIP_CHECK(hipMemsetAsync(scratch.counter, 0, 4, s));
    const size_t host_tail = sumcheck_host_tail_len();
    std::vector<std::vector<Fe<F>>> host_tabs;  // ninst x np tables once the rounds have moved to the host (sumcheck_host_round)
    std::vector<size_t> host_len;
    for (size_t m = n; m > 1; m /= 2, j++) {
        Fe<F> ev[3] = {fe_zero<F>(), fe_zero<F>(), fe_zero<F>()};
        if (host_tabs.empty() && host_tail && length <= host_tail) {
            host_tabs.resize(ninst * (size_t)np);
            host_len.assign(ninst, length);
            for (size_t k = 0; k < ninst * (size_t)np; k++) {
                host_tabs[k].resize(length);
                LURK_HIP_CHECK(hipMemcpyAsync(host_tabs[k].data(), d_polys[k], length * 32, hipMemcpyDeviceToHost, s));
            }
            LURK_HIP_CHECK(hipStreamSynchronize(s));
        }
        for (size_t i = 0; i < ninst; i++) {
            Fe<F> one_ev[3];
            if (!host_tabs.empty()) sumcheck_host_round<F>(np, host_tabs.data() + i * np, host_len[i], have_r ? &r : nullptr, one_ev);
            else sumcheck_round<F>(np, d_polys + i * np, length, have_r ? (const void*)r.l : nullptr, one_ev, s, &scratch);  // Montgomery images of the evaluations at 0, 2 (, 3)
            for (int k = 0; k < nv; k++) ev[k] = ninst == 1 && !coeffs32_canonical ? one_ev[k] : fe_add<F>(ev[k], fe_mul<F>(coeff[i], one_ev[k]));
        }
        if (have_r) length /= 2;
        const Fe<F> e0 = ev[0], e2 = ev[1], e1 = fe_sub<F>(claim, e0);
        Fe<F> poly[4];
        const Fe<F> second = fe_add<F>(fe_sub<F>(e2, fe_mul<F>(two, e1)), e0);  // e2 - 2 e1 + e0
        if (cubic) {
            const Fe<F> e3 = ev[2];
            const Fe<F> a3 = fe_mul<F>(fe_sub<F>(fe_add<F>(fe_sub<F>(e3, fe_mul<F>(three, e2)), fe_mul<F>(three, e1)), e0), inv6);
            const Fe<F> b = fe_sub<F>(fe_mul<F>(second, inv2), fe_mul<F>(three, a3));
            poly[0] = e0;
            poly[1] = fe_sub<F>(fe_sub<F>(fe_sub<F>(e1, e0), a3), b);
            poly[2] = b;
            poly[3] = a3;
        } else {
            const Fe<F> a2 = fe_mul<F>(second, inv2);
            poly[0] = e0;
            poly[1] = fe_sub<F>(fe_sub<F>(e1, e0), a2);
            poly[2] = a2;
        }
        uint64_t* out = out_polys + (size_t)j * ncoef * 4;
        for (int k = 0; k < ncoef; k++) {
            const Fe<F> c = fe_from_mont<F>(poly[k]);
            memcpy(out + 4 * k, c.l, 32);
        }
        uint64_t r_can[4] = {0, 0, 0, 0};
        LURK_REQUIRE(challenge(user, j, out, r_can) == 0, "the transcript callback failed");
        memcpy(r.l, r_can, 32);
        LURK_REQUIRE(!fe_canonical_ge_mod<F>(r.l), "the challenge is not reduced modulo the field order");
        r = fe_to_mont<F>(r);
        have_r = true;
        Fe<F> acc = fe_zero<F>();
        for (int k = ncoef - 1; k >= 0; k--) acc = fe_add<F>(fe_mul<F>(acc, r), poly[k]);
        claim = acc;
    }
    if (!host_tabs.empty()) {
        // the last bind and the final evaluations, on the host tables
        for (size_t k = 0; k < ninst * (size_t)np; k++) {
            const std::vector<Fe<F>>& t = host_tabs[k];
            const Fe<F> v = fe_from_mont<F>(have_r && t.size() >= 2 ? sc_bind<F>(t[0], t[1], r) : t[0]);
            memcpy(out_finals + 4 * k, v.l, 32);
        }
    } else {
        for (size_t i = 0; have_r && i < ninst; i++) sumcheck_round<F>(np, d_polys + i * np, length, r.l, nullptr, s, &scratch);  // the last bind: every table is down to one element
        for (size_t k = 0; k < ninst * (size_t)np; k++) {
            Fe<F> v;
            LURK_HIP_CHECK(hipMemcpyAsync(v.l, d_polys[k], 32, hipMemcpyDeviceToHost, s));
            LURK_HIP_CHECK(hipStreamSynchronize(s));
            v = fe_from_mont<F>(v);
            memcpy(out_finals + 4 * k, v.l, 32);
        }
    }
    claim = fe_from_mont<F>(claim);
    memcpy(out_claim32, claim.l, 32);
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_sumcheck_round_dev(int field_id, int degree, void* const* d_polys, size_t len, const void* bind_r32_mont, void* evals_out,
                                void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(degree == 2 || degree == 3, "degree must be 2 (a b) or 3 (a (b c - d))");
        LURK_REQUIRE(d_polys, "null table list");
        const int np = degree == 3 ? 4 : 2;
        for (int k = 0; k < np; k++) LURK_REQUIRE(d_polys[k], "null table");
        LURK_REQUIRE(evals_out || bind_r32_mont, "nothing to do");
        LURK_REQUIRE(evals_out == nullptr || bind_r32_mont == nullptr || len >= 4, "after the last bind (len = 2) there is nothing to sum: pass evals_out = NULL");
        if (field_id == 0) sumcheck_round<PallasFp>(np, d_polys, len, bind_r32_mont, evals_out, (hipStream_t)stream);
        else if (field_id == 1) sumcheck_round<PallasFq>(np, d_polys, len, bind_r32_mont, evals_out, (hipStream_t)stream);
        else sumcheck_round<Bn254Fr>(np, d_polys, len, bind_r32_mont, evals_out, (hipStream_t)stream);
    });
}

int lurk_hip_sumcheck_prove_dev(int field_id, int degree, void* const* d_polys, size_t len, const void* claim32_canonical,
                                lurk_hip_sumcheck_challenge_fn challenge, void* user, void* out_polys, void* out_finals, void* out_claim32, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(degree == 2 || degree == 3, "degree must be 2 (a b) or 3 (a (b c - d))");
        LURK_REQUIRE(d_polys && claim32_canonical && challenge && out_polys && out_finals && out_claim32, "null argument");
        LURK_REQUIRE(len >= 2 && (len & (len - 1)) == 0, "table length must be a power of two >= 2");
        const int np = degree == 3 ? 4 : 2;
        for (int k = 0; k < np; k++) LURK_REQUIRE(d_polys[k], "null table");
        if (field_id == 0) sumcheck_prove<PallasFp>(np, 1, d_polys, len, nullptr, claim32_canonical, challenge, user, (uint64_t*)out_polys, (uint64_t*)out_finals, out_claim32, (hipStream_t)stream);
        else if (field_id == 1) sumcheck_prove<PallasFq>(np, 1, d_polys, len, nullptr, claim32_canonical, challenge, user, (uint64_t*)out_polys, (uint64_t*)out_finals, out_claim32, (hipStream_t)stream);
        else sumcheck_prove<Bn254Fr>(np, 1, d_polys, len, nullptr, claim32_canonical, challenge, user, (uint64_t*)out_polys, (uint64_t*)out_finals, out_claim32, (hipStream_t)stream);
    });
}

int lurk_hip_sumcheck_prove_batch_dev(int field_id, int degree, size_t n_instances, void* const* d_polys, size_t len, const void* coeffs32_canonical,
                                      const void* claim32_canonical, lurk_hip_sumcheck_challenge_fn challenge, void* user, void* out_polys,
                                      void* out_finals, void* out_claim32, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(degree == 2 || degree == 3, "degree must be 2 (a b) or 3 (a (b c - d))");
        LURK_REQUIRE(n_instances >= 1 && n_instances <= 4096, "instance count out of range");
        LURK_REQUIRE(d_polys && coeffs32_canonical && claim32_canonical && challenge && out_polys && out_finals && out_claim32, "null argument");
        LURK_REQUIRE(len >= 2 && (len & (len - 1)) == 0, "table length must be a power of two >= 2");
        const int np = degree == 3 ? 4 : 2;
        for (size_t k = 0; k < n_instances * (size_t)np; k++) LURK_REQUIRE(d_polys[k], "null table");
        if (field_id == 0) sumcheck_prove<PallasFp>(np, n_instances, d_polys, len, coeffs32_canonical, claim32_canonical, challenge, user, (uint64_t*)out_polys, (uint64_t*)out_finals, out_claim32, (hipStream_t)stream);
        else if (field_id == 1) sumcheck_prove<PallasFq>(np, n_instances, d_polys, len, coeffs32_canonical, claim32_canonical, challenge, user, (uint64_t*)out_polys, (uint64_t*)out_finals, out_claim32, (hipStream_t)stream);
        else sumcheck_prove<Bn254Fr>(np, n_instances, d_polys, len, coeffs32_canonical, claim32_canonical, challenge, user, (uint64_t*)out_polys, (uint64_t*)out_finals, out_claim32, (hipStream_t)stream);
    });
}

int lurk_hip_eq_evals_dev(int field_id, const void* r32_mont, int ell, void* d_out, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(ell >= 0 && ell <= 30 && d_out && (ell == 0 || r32_mont), "bad argument");
        hipStream_t s = (hipStream_t)stream;
        ArenaBuf r_buf((size_t)(ell ? ell : 1) * 32, s);
        void* d_r = r_buf.p;
        if (ell) LURK_HIP_CHECK(hipMemcpyAsync(d_r, r32_mont, (size_t)ell * 32, hipMemcpyHostToDevice, s));
        const size_t n = (size_t)1 << ell;
        const int lo = ell < EQ_LO_BITS ? ell : EQ_LO_BITS;
        unsigned blocks = (unsigned)(n >> lo), cap = (unsigned)num_cus() * 4;  // one chunk of 2^lo outputs per workgroup and pass
        if (blocks > cap) blocks = cap;
        const size_t lds = ((size_t)1 << lo) * 32;
        ProfScope ps("eq_evals", s);
        if (field_id == 0) hipLaunchKernelGGL((eq_evals_kernel<PallasFp>), dim3(blocks), dim3(SC_BLOCK), lds, s, (const Fe<PallasFp>*)d_r, ell, (Fe<PallasFp>*)d_out);
        else if (field_id == 1) hipLaunchKernelGGL((eq_evals_kernel<PallasFq>), dim3(blocks), dim3(SC_BLOCK), lds, s, (const Fe<PallasFq>*)d_r, ell, (Fe<PallasFq>*)d_out);
        else hipLaunchKernelGGL((eq_evals_kernel<Bn254Fr>), dim3(blocks), dim3(SC_BLOCK), lds, s, (const Fe<Bn254Fr>*)d_r, ell, (Fe<Bn254Fr>*)d_out);
        LURK_HIP_CHECK(hipGetLastError());
    });
}
}

// transcript.hip - the folding challenge r: Nova's random oracle (host code; no kernel in this file).
//
// Reference: the closure lurk-beta hands every step to (/root/reference/src/proof/nova.rs:282-295) runs arecibo's
// RecursiveSNARK::prove_step -> NIFS::prove, which derives r between commit(T) and the fold:
//     ro = RO::new(ro_consts, NUM_FE_FOR_RO); ro.absorb(pp_digest); U1.absorb_in_ro(ro); U2.absorb_in_ro(ro);
//     comm_T.absorb_in_ro(ro); r = ro.squeeze(NUM_CHALLENGE_BITS)
// RO = PoseidonRO<Base, Scalar> over neptune's sponge API.  arecibo and neptune are un-vendored git
// dependencies (/root/reference/Cargo.toml:127-128), so everything below is restated from their published sources [MEM] and is
// UNPINNED: no transcript value exists in /root/reference.  What is restated:
//   * neptune Sponge::api_constants(Strength::Standard), arity U24: width t = 25, the round numbers / Grain-LFSR constants /
//     Cauchy MDS of poseidon_params.hpp (the same generator as the hash3/4/6/8 constants, which ARE pinned by the KATs);
//   * neptune sponge API, simplex mode: capacity element state[0] = the IO-pattern tag, rate = 24 elements state[1..25);
//     absorb adds into the rate (a permutation when the rate is full), squeeze permutes when nothing is left to read;
//     IOPattern([Absorb(n), Squeeze(1)]).value(domain separator 0): x = 2^128 - 159, state = sum_k x^k * op_k mod 2^128 with
//     op = n + 2^31 for an absorb and n for a squeeze, the domain separator as the last term;
//   * PoseidonRO::squeeze: the low num_bits bits (little-endian) of the squeezed element, read as a scalar;
//   * absorb_in_ro: a commitment is (x, y, is_infinity) of its affine form (identity: 0, 0, 1); a relaxed instance is
//     comm_W, comm_E, u, then every X_i as BN_N_LIMBS = 4 limbs of BN_LIMB_WIDTH = 64 bits; a fresh instance is comm_W, X_i;
//     scalars enter the base field through their canonical integer (scalar_as_base).
// The permutation runs the sparse schedule of poseidon_params.hpp on 4 x 64-bit host limbs with one reduction per matrix row.
#include <map>
#include <memory>
#include <mutex>

#include "common.hpp"
#include "host_poseidon.hpp"
#include "keccak.hpp"

namespace lurk {

// ---- the run-time parameters of this restatement (round 6) -----------------------------------------------------------------------
// Everything above that is recalled from memory [MEM] and cannot be checked against /root/reference is a FIELD of lurk_hip_ro_params
// (include/lurk_hip.h), process-wide, defaults = what rounds 1-5 compiled in: a Rust host whose first r differs from arecibo's can
// move one field at a time (or let lurk_beta_amd/dump.py search them against a LURKDUMP probe record) instead of rebuilding the
// library.  A folding context snapshots the parameters when a step's transcript begins.
static lurk_hip_ro_params ro_params_default() {
    lurk_hip_ro_params p;
    memset(&p, 0, sizeof(p));
    p.struct_size = (uint32_t)sizeof(p);
    p.arity = 24;               // neptune U24: rate of the sponge arecibo's PoseidonRO is built on
    p.domain_separator = 0;
    p.absorb_tag_bit = 31;      // SpongeOp::Absorb(n).value() = n + 2^31
    p.num_challenge_bits = 128; // NUM_CHALLENGE_BITS
    for (uint32_t i = 0; i < 4; i++) p.item_order[i] = i;     // pp_digest, U1, U2, comm_T
    for (uint32_t i = 0; i < 4; i++) p.relaxed_order[i] = i;  // comm_W, comm_E, u, X
    for (uint32_t i = 0; i < 2; i++) p.fresh_order[i] = i;    // comm_W, X
    p.point_elements = 3;       // (x, y, is_infinity)
    p.relaxed_x_limbs = 4;      // BN_N_LIMBS
    p.fresh_x_limbs = 0;        // one element through scalar_as_base
    p.limb_bits = 64;           // BN_LIMB_WIDTH
    p.pattern_absorbs = 0;      // the IO pattern declares what is absorbed
    p.squeeze_element = 0;
    return p;
}
static std::mutex g_ro_mu;
static lurk_hip_ro_params g_ro = ro_params_default();
static lurk_hip_ro_params ro_params() {
    std::lock_guard<std::mutex> lk(g_ro_mu);
    return g_ro;
}
static bool is_permutation(const uint32_t* v, uint32_t n) {
    uint32_t seen = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (v[i] >= n || (seen >> v[i] & 1)) return false;
        seen |= 1u << v[i];
    }
    return true;
}
static void ro_params_validate(const lurk_hip_ro_params& p) {
    LURK_REQUIRE(p.struct_size == sizeof(lurk_hip_ro_params), "lurk_hip_ro_params: struct_size does not match this library's layout");
    LURK_REQUIRE(p.arity >= 2 && p.arity <= 36, "lurk_hip_ro_params: arity must be in 2..36");
    LURK_REQUIRE(p.absorb_tag_bit <= 31, "lurk_hip_ro_params: absorb_tag_bit must be in 0..31");
    LURK_REQUIRE(p.num_challenge_bits >= 1 && p.num_challenge_bits <= 250, "lurk_hip_ro_params: num_challenge_bits must be in 1..250");
    LURK_REQUIRE(is_permutation(p.item_order, 4), "lurk_hip_ro_params: item_order is not a permutation of 0..3");
    LURK_REQUIRE(is_permutation(p.relaxed_order, 4), "lurk_hip_ro_params: relaxed_order is not a permutation of 0..3");
    LURK_REQUIRE(is_permutation(p.fresh_order, 2), "lurk_hip_ro_params: fresh_order is not a permutation of 0..1");
    LURK_REQUIRE(p.point_elements == 2 || p.point_elements == 3, "lurk_hip_ro_params: point_elements must be 2 or 3");
    LURK_REQUIRE(p.relaxed_x_limbs <= 16 && p.fresh_x_limbs <= 16, "lurk_hip_ro_params: at most 16 limbs per element");
    LURK_REQUIRE(p.limb_bits >= 1 && p.limb_bits <= 250, "lurk_hip_ro_params: limb_bits must be in 1..250");
    LURK_REQUIRE(p.pattern_absorbs < (1u << 31), "lurk_hip_ro_params: pattern_absorbs out of range");
    LURK_REQUIRE(p.squeeze_element < p.arity, "lurk_hip_ro_params: squeeze_element must be below the arity");
}

template <class P>
static const RoHost& ro_host(uint32_t arity) {
    return poseidon_host<P>((int)arity);  // one table per (field, arity), built on first use
}

// neptune sponge/api.rs: IOPattern::value
static unsigned __int128 io_pattern_tag(uint32_t absorbs, uint32_t squeezes, uint32_t domain_separator, uint32_t absorb_tag_bit = 31) {
    const unsigned __int128 x = (unsigned __int128)0 - 159;
    unsigned __int128 xi = 1, st = 0;
    auto update = [&](uint32_t a) {
        xi *= x;
        st += xi * a;
    };
    if (absorbs) update(absorbs + (1u << absorb_tag_bit));
    if (squeezes) update(squeezes);
    update(domain_separator);
    return st;
}

// The sponge as a resumable object: the IO pattern (hence the capacity tag) is fixed by the TOTAL absorb count, elements may then arrive
// in any number of calls; a permutation runs whenever the rate is full and one more element arrives.  This is what lets the folding
// context absorb the running instance (and run the permutation it completes) while the device is still computing comm_T.
template <class P>
struct RoSponge {
    const RoHost* R = nullptr;
    std::vector<H4> s;
    int pos = 0;
    size_t total = 0, absorbed = 0;
    uint32_t squeeze_element = 0;
    void begin(size_t n, const lurk_hip_ro_params& prm) {
        R = &ro_host<P>(prm.arity);
        squeeze_element = prm.squeeze_element;
        const H4 zero = {{0, 0, 0, 0}};
        s.assign(R->t, zero);
        const unsigned __int128 tag = io_pattern_tag(prm.pattern_absorbs ? prm.pattern_absorbs : (uint32_t)n, 1, prm.domain_separator, prm.absorb_tag_bit);
        H4 c = {{(uint64_t)tag, (uint64_t)(tag >> 64), 0, 0}};
        s[0] = h4_mul(R->F, c, R->r2);
        pos = 0;
        total = n;
        absorbed = 0;
    }
    void absorb(const uint64_t* elems, size_t k) {
        const int rate = R->t - 1;
        LURK_REQUIRE(absorbed + k <= total, "random oracle: more elements than the declared IO pattern");
        for (size_t i = 0; i < k; i++) {
            if (pos == rate) {
                permute_host(*R, s);
                pos = 0;
            }
            H4 e;
            memcpy(e.v, elems + 4 * i, 32);
            LURK_REQUIRE(!h4_geq(e.v, R->F.m), "random oracle input is not a canonical field element");
            s[1 + pos] = h4_add(R->F, s[1 + pos], h4_mul(R->F, e, R->r2));
            pos++;
        }
        absorbed += k;
    }
    // a permutation that only waits for the NEXT element to arrive can run now (the element then lands in a fresh rate)
    void permute_if_full() {
        if (pos == R->t - 1 && absorbed < total) {
            permute_host(*R, s);
            pos = 0;
        }
    }
    Fe<P> squeeze() {  // squeeze position = rate after an absorb: one permutation, then read rate element 0
        LURK_REQUIRE(absorbed == total, "random oracle: fewer elements than the declared IO pattern");
        const H4 one = {{1, 0, 0, 0}};
        permute_host(*R, s);
        return h4_to<P>(h4_mul(R->F, s[1 + squeeze_element], one));
    }
};

// absorb n canonical elements, squeeze one; returns it in canonical form
template <class P>
static Fe<P> ro_squeeze_host(const uint64_t* elems, size_t n) {
    RoSponge<P> sp;
    sp.begin(n, ro_params());
    sp.absorb(elems, n);
    return sp.squeeze();
}

static void ro_squeeze(int field_id, const uint64_t* elems, size_t n, unsigned num_bits, uint64_t* out4) {
    LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
    LURK_REQUIRE(num_bits >= 1 && num_bits <= 250, "num_bits must be in 1..250");
    LURK_REQUIRE(n >= 1 && n < ((size_t)1 << 31), "absorb count out of range");
    uint32_t w[8];
    if (field_id == 0) memcpy(w, ro_squeeze_host<PallasFp>(elems, n).l, 32);
    else if (field_id == 1) memcpy(w, ro_squeeze_host<PallasFq>(elems, n).l, 32);
    else memcpy(w, ro_squeeze_host<Bn254Fr>(elems, n).l, 32);
    for (unsigned b = num_bits; b < 256; b++) w[b >> 5] &= ~(1u << (b & 31));
    memcpy(out4, w, 32);
}

// canonical integer of a Montgomery element of field F, reduced into field B (scalar_as_base: Fq is the larger Pasta field)
template <class F, class B>
static void scalar_as_base(const void* x_mont, uint64_t* out4) {
    Fe<F> x;
    memcpy(x.l, x_mont, 32);
    x = fe_from_mont<F>(x);
    uint32_t w[8];
    memcpy(w, x.l, 32);
    if (fe_canonical_ge_mod<B>(w)) {  // at most one subtraction: both moduli are 2^254 + eps
        uint32_t borrow = 0;
        for (int i = 0; i < 8; i++) w[i] = subb32(w[i], B::mod(i), borrow);
    }
    memcpy(out4, w, 32);
}
// n_limbs limbs of limb_bits bits of the canonical integer, low limb first (BN_N_LIMBS = 4 limbs of BN_LIMB_WIDTH = 64 bits by default)
template <class F>
static void scalar_limbs(const void* x_mont, uint32_t n_limbs, uint32_t limb_bits, std::vector<uint64_t>& out) {
    Fe<F> x;
    memcpy(x.l, x_mont, 32);
    x = fe_from_mont<F>(x);
    auto bit = [&](uint32_t b) -> uint64_t { return b < 256 ? (x.l[b >> 5] >> (b & 31)) & 1u : 0u; };
    for (uint32_t k = 0; k < n_limbs; k++) {
        uint64_t w[4] = {0, 0, 0, 0};
        for (uint32_t b = 0; b < limb_bits; b++) w[b >> 6] |= bit(k * limb_bits + b) << (b & 63);
        out.insert(out.end(), w, w + 4);
    }
}
static void absorb_commitment(int curve, const void* jac96, uint32_t point_elements, std::vector<uint64_t>& out) {
    uint64_t xy[8];
    if (lurk_hip_point_to_affine_canonical(curve, xy, jac96) != 0) throw HipFailure{LURK_HIP_ERR_INVALID_ARG, lurk_hip_last_error()};
    bool inf = true;
    for (int i = 0; i < 8; i++) inf = inf && xy[i] == 0;
    out.insert(out.end(), xy, xy + 8);
    if (point_elements == 3) {
        out.push_back(inf ? 1 : 0);
        out.push_back(0);
        out.push_back(0);
        out.push_back(0);
    }
}

// F = scalar field of `curve`, B = its base field (the RO's field).  In three stages, so that a folding context can run the first two
// while the device is still busy with the step (nifs_pre.hpp): `begin` brings pp_digest and U1, `fresh` U2 = (comm_W2, X2), `finish`
// comm_T and squeezes.  The four items are absorbed in lurk_hip_ro_params::item_order as soon as every item before them has arrived:
// in the default order a Lurk step (num_io = 6: 44 elements in all) fills the first rate of 24 inside `begin` and its permutation runs
// at once; one permutation and one affine conversion are all that is left behind commit(T).
template <class F, class B>
struct NifsStages {
    RoSponge<B> sp;
    lurk_hip_ro_params prm;
    int curve = 0;
    size_t num_io = 0;
    std::vector<uint64_t> item[4];
    bool have[4] = {false, false, false, false};
    int next = 0;
    void drain() {
        while (next < 4 && have[prm.item_order[next]]) {
            std::vector<uint64_t>& el = item[prm.item_order[next]];
            sp.absorb(el.data(), el.size() / 4);
            sp.permute_if_full();
            next++;
        }
    }
    void x_elements(const void* x_mont, uint32_t limbs, std::vector<uint64_t>& el) {
        uint64_t tmp[4];
        for (size_t i = 0; i < num_io; i++) {
            if (limbs) {
                scalar_limbs<F>((const char*)x_mont + 32 * i, limbs, prm.limb_bits, el);
            } else {
                scalar_as_base<F, B>((const char*)x_mont + 32 * i, tmp);
                el.insert(el.end(), tmp, tmp + 4);
            }
        }
    }
    void begin(int curve_, const void* pp_digest32, const void* comm_w1, const void* comm_e1, const void* u1_mont, const void* x1_mont, size_t num_io_) {
        prm = ro_params();
        curve = curve_;
        num_io = num_io_;
        {
            // the digest is a scalar's canonical bytes: scalar_as_base through a Montgomery round trip is not needed, reduce directly
            uint32_t w[8];
            memcpy(w, pp_digest32, 32);
            LURK_REQUIRE(!fe_canonical_ge_mod<F>(w), "pp_digest is not a canonical scalar");
            if (fe_canonical_ge_mod<B>(w)) {
                uint32_t borrow = 0;
                for (int i = 0; i < 8; i++) w[i] = subb32(w[i], B::mod(i), borrow);
            }
            uint64_t d[4];
            memcpy(d, w, 32);
            item[LURK_RO_ITEM_PP_DIGEST].assign(d, d + 4);
            have[LURK_RO_ITEM_PP_DIGEST] = true;
        }
        std::vector<uint64_t>& el = item[LURK_RO_ITEM_U1];  // U1: RelaxedR1CSInstance::absorb_in_ro
        el.clear();
        for (int k = 0; k < 4; k++) {
            switch (prm.relaxed_order[k]) {
                case LURK_RO_PART_COMM_W: absorb_commitment(curve, comm_w1, prm.point_elements, el); break;
                case LURK_RO_PART_COMM_E: absorb_commitment(curve, comm_e1, prm.point_elements, el); break;
                case LURK_RO_PART_U: {
                    uint64_t tmp[4];
                    scalar_as_base<F, B>(u1_mont, tmp);
                    el.insert(el.end(), tmp, tmp + 4);
                    break;
                }
                default: x_elements(x1_mont, prm.relaxed_x_limbs, el); break;
            }
        }
        have[LURK_RO_ITEM_U1] = true;
        const size_t pt = prm.point_elements;
        sp.begin(1 + (2 * pt + 1 + num_io * (prm.relaxed_x_limbs ? prm.relaxed_x_limbs : 1)) + (pt + num_io * (prm.fresh_x_limbs ? prm.fresh_x_limbs : 1)) + pt, prm);
        drain();
    }
    void fresh(const void* comm_w2, const void* x2_mont) {
        std::vector<uint64_t>& el = item[LURK_RO_ITEM_U2];  // U2: R1CSInstance::absorb_in_ro
        el.clear();
        for (int k = 0; k < 2; k++) {
            if (prm.fresh_order[k] == 0) absorb_commitment(curve, comm_w2, prm.point_elements, el);
            else x_elements(x2_mont, prm.fresh_x_limbs, el);
        }
        have[LURK_RO_ITEM_U2] = true;
        drain();
    }
    void finish(const void* comm_t, void* r32_mont) {
        item[LURK_RO_ITEM_COMM_T].clear();
        absorb_commitment(curve, comm_t, prm.point_elements, item[LURK_RO_ITEM_COMM_T]);
        have[LURK_RO_ITEM_COMM_T] = true;
        drain();
        LURK_REQUIRE(next == 4, "the transcript is missing an item (begin / fresh / finish out of order)");
        Fe<B> sq = sp.squeeze();
        uint32_t w[8];
        memcpy(w, sq.l, 32);
        for (unsigned b = prm.num_challenge_bits; b < 256; b++) w[b >> 5] &= ~(1u << (b & 31));  // NUM_CHALLENGE_BITS
        Fe<F> rf;
        memcpy(rf.l, w, 32);
        while (fe_canonical_ge_mod<F>(rf.l)) {  // only with more than 253 challenge bits: the integer is read as a scalar
            uint32_t borrow = 0;
            for (int i = 0; i < 8; i++) rf.l[i] = subb32(rf.l[i], F::mod(i), borrow);
        }
        rf = fe_to_mont<F>(rf);
        memcpy(r32_mont, rf.l, 32);
    }
    // the elements in absorb order (diagnostics: lurk_hip_nifs_absorb_list); valid once all four items are present
    std::vector<uint64_t> absorb_list() const {
        std::vector<uint64_t> out;
        for (int k = 0; k < 4; k++) out.insert(out.end(), item[prm.item_order[k]].begin(), item[prm.item_order[k]].end());
        return out;
    }
};
template <class F, class B>
static void nifs_challenge(int curve, int /*base_field_id*/, const void* pp_digest32, const void* comm_w1, const void* comm_e1, const void* u1_mont,
                           const void* x1_mont, const void* comm_w2, const void* x2_mont, size_t num_io, const void* comm_t, void* r32_mont) {
    NifsStages<F, B> st;
    st.begin(curve, pp_digest32, comm_w1, comm_e1, u1_mont, x1_mont, num_io);
    st.fresh(comm_w2, x2_mont);
    st.finish(comm_t, r32_mont);
}

// the staged form behind an opaque handle (nifs_pre.hpp: step.hip holds one per open step)
struct NifsPre {
    int curve = 0;
    NifsStages<PallasFq, PallasFp> pallas;  // curve 0: scalars Fq, the RO over Fp
    NifsStages<PallasFp, PallasFq> vesta;
};
NifsPre* nifs_pre_begin(int curve, const void* pp_digest32, const void* comm_w1, const void* comm_e1, const void* u1_mont, const void* x1_mont, size_t num_io) {
    LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
    auto p = std::make_unique<NifsPre>();
    p->curve = curve;
    if (curve == LURK_CURVE_PALLAS) p->pallas.begin(curve, pp_digest32, comm_w1, comm_e1, u1_mont, x1_mont, num_io);
    else p->vesta.begin(curve, pp_digest32, comm_w1, comm_e1, u1_mont, x1_mont, num_io);
    return p.release();
}
void nifs_pre_fresh(NifsPre* p, const void* comm_w2, const void* x2_mont) {
    if (p->curve == LURK_CURVE_PALLAS) p->pallas.fresh(comm_w2, x2_mont);
    else p->vesta.fresh(comm_w2, x2_mont);
}
void nifs_pre_finish(NifsPre* p, const void* comm_t, void* r32_mont) {
    if (p->curve == LURK_CURVE_PALLAS) p->pallas.finish(comm_t, r32_mont);
    else p->vesta.finish(comm_t, r32_mont);
}
void nifs_pre_free(NifsPre* p) { delete p; }

// ---- arecibo's Keccak256Transcript (the transcript of RelaxedR1CSSNARK / BatchedRelaxedR1CSSNARK behind CompressedSNARK::prove,
// /root/reference/src/proof/nova.rs:92, 341-356, supernova.rs:110, 293-302) -----------------------------------------------------------
// Restated from arecibo's published source [MEM] and UNPINNED, like the random oracle above (oracle/keccak_transcript.py spells the
// construction out; Keccak-256 itself is pinned by known answers).  64-byte state, u16 round counter, a running Keccak256 hasher.
static void keccak_updated_state(const Keccak256& h, const void* input, size_t len, uint8_t out[64]) {
    Keccak256 k = h;
    k.update(input, len);
    Keccak256 lo = k, hi = k;
    const uint8_t zero = 0, one = 1;
    lo.update(&zero, 1);
    hi.update(&one, 1);
    lo.finalize(out);
    hi.finalize(out + 32);
}
// Scalar::from_uniform: the 64 bytes as a little-endian integer, reduced
template <class P>
static void from_uniform64(const uint8_t in[64], uint64_t out4[4]) {
    const HostField F = host_field<P>();
    H4 lo, hi;
    memcpy(lo.v, in, 32);
    memcpy(hi.v, in + 32, 32);
    while (h4_geq(lo.v, F.m)) h4_sub_m(lo.v, F.m);
    while (h4_geq(hi.v, F.m)) h4_sub_m(hi.v, F.m);
    const H4 r2 = h4_from<P>(fe_r2<P>());
    const H4 hi_shift = h4_mul(F, hi, r2);  // hi * 2^512 / 2^256 = hi * 2^256
    const H4 v = h4_add(F, lo, hi_shift);
    memcpy(out4, v.v, 32);
}

}  // namespace lurk

struct lurk_hip_keccak_transcript {
    uint16_t round = 0;
    uint8_t state[64];
    lurk::Keccak256 hasher;
};

using namespace lurk;

// host-only entry points: no device is needed (the CPU tests compare them with oracle/pyref.py)
template <class F>
static int host_guarded(F&& f) {
    try {
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

extern "C" {

int lurk_hip_nova_ro_squeeze(int field_id, const void* elems32, size_t n, unsigned num_bits, void* out32) {
    return host_guarded([&] {
        LURK_REQUIRE(elems32 && out32, "null argument");
        std::vector<uint64_t> in(4 * n);
        memcpy(in.data(), elems32, 32 * n);
        uint64_t r[4];
        ro_squeeze(field_id, in.data(), n, num_bits, r);
        memcpy(out32, r, 32);
    });
}

int lurk_hip_nova_ro_pattern_tag(uint32_t absorbs, uint32_t squeezes, uint32_t domain_separator, void* out16) {
    return host_guarded([&] {
        LURK_REQUIRE(out16, "null argument");
        const unsigned __int128 t = io_pattern_tag(absorbs, squeezes, domain_separator, ro_params().absorb_tag_bit);
        memcpy(out16, &t, 16);
    });
}

int lurk_hip_ro_params_get(lurk_hip_ro_params* out) {
    return host_guarded([&] {
        LURK_REQUIRE(out, "null argument");
        *out = ro_params();
    });
}
int lurk_hip_ro_params_set(const lurk_hip_ro_params* params) {
    return host_guarded([&] {
        const lurk_hip_ro_params p = params ? *params : ro_params_default();  // NULL: back to the defaults
        ro_params_validate(p);
        std::lock_guard<std::mutex> lk(g_ro_mu);
        g_ro = p;
    });
}

int lurk_hip_nifs_absorb_list(int curve, const void* pp_digest32, const void* comm_w1_jac96, const void* comm_e1_jac96, const void* u1_mont,
                              const void* x1_mont, const void* comm_w2_jac96, const void* x2_mont, size_t num_io, const void* comm_t_jac96,
                              void* out_elems32, size_t cap, size_t* count) {
    return host_guarded([&] {
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE(pp_digest32 && comm_w1_jac96 && comm_e1_jac96 && u1_mont && comm_w2_jac96 && comm_t_jac96 && count, "null argument");
        LURK_REQUIRE(num_io == 0 || (x1_mont && x2_mont), "null public IO");
        std::vector<uint64_t> el;
        uint64_t r[4];
        if (curve == LURK_CURVE_PALLAS) {
            NifsStages<PallasFq, PallasFp> st;
            st.begin(curve, pp_digest32, comm_w1_jac96, comm_e1_jac96, u1_mont, x1_mont, num_io);
            st.fresh(comm_w2_jac96, x2_mont);
            st.finish(comm_t_jac96, r);
            el = st.absorb_list();
        } else {
            NifsStages<PallasFp, PallasFq> st;
            st.begin(curve, pp_digest32, comm_w1_jac96, comm_e1_jac96, u1_mont, x1_mont, num_io);
            st.fresh(comm_w2_jac96, x2_mont);
            st.finish(comm_t_jac96, r);
            el = st.absorb_list();
        }
        *count = el.size() / 4;
        if (out_elems32) {
            LURK_REQUIRE(cap >= *count, "the output buffer is shorter than the absorb list");
            memcpy(out_elems32, el.data(), el.size() * 8);
        }
    });
}

int lurk_hip_nifs_challenge(int curve, const void* pp_digest32, const void* comm_w1_jac96, const void* comm_e1_jac96, const void* u1_mont,
                            const void* x1_mont, const void* comm_w2_jac96, const void* x2_mont, size_t num_io, const void* comm_t_jac96,
                            void* r32_mont) {
    return host_guarded([&] {
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE(pp_digest32 && comm_w1_jac96 && comm_e1_jac96 && u1_mont && comm_w2_jac96 && comm_t_jac96 && r32_mont, "null argument");
        LURK_REQUIRE(num_io == 0 || (x1_mont && x2_mont), "null public IO");
        if (curve == LURK_CURVE_PALLAS)
            nifs_challenge<PallasFq, PallasFp>(curve, LURK_FIELD_PALLAS_FP, pp_digest32, comm_w1_jac96, comm_e1_jac96, u1_mont, x1_mont, comm_w2_jac96,
                                               x2_mont, num_io, comm_t_jac96, r32_mont);
        else
            nifs_challenge<PallasFp, PallasFq>(curve, LURK_FIELD_PALLAS_FQ, pp_digest32, comm_w1_jac96, comm_e1_jac96, u1_mont, x1_mont, comm_w2_jac96,
                                               x2_mont, num_io, comm_t_jac96, r32_mont);
    });
}

int lurk_hip_keccak256(const void* in, size_t len, void* out32) {
    return host_guarded([&] {
        LURK_REQUIRE((in || len == 0) && out32, "null argument");
        Keccak256 k;
        k.update(in, len);
        k.finalize((uint8_t*)out32);
    });
}

int lurk_hip_keccak_transcript_new(lurk_hip_keccak_transcript** t, const void* label, size_t label_len) {
    return host_guarded([&] {
        LURK_REQUIRE(t && (label || label_len == 0), "null argument");
        auto tr = std::make_unique<lurk_hip_keccak_transcript>();
        std::vector<uint8_t> in = {'N', 'o', 'T', 'R'};  // PERSONA_TAG
        in.insert(in.end(), (const uint8_t*)label, (const uint8_t*)label + label_len);
        keccak_updated_state(Keccak256(), in.data(), in.size(), tr->state);
        *t = tr.release();
    });
}
int lurk_hip_keccak_transcript_destroy(lurk_hip_keccak_transcript* t) {
    delete t;
    return 0;
}
int lurk_hip_keccak_transcript_absorb(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, const void* bytes, size_t len) {
    return host_guarded([&] {
        LURK_REQUIRE(t && (label || label_len == 0) && (bytes || len == 0), "null argument");
        t->hasher.update(label, label_len);
        t->hasher.update(bytes, len);
    });
}
int lurk_hip_keccak_transcript_absorb_scalars(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, const void* scalars32_canonical, size_t n) {
    return host_guarded([&] {
        LURK_REQUIRE(t && (label || label_len == 0) && (scalars32_canonical || n == 0), "null argument");
        t->hasher.update(label, label_len);
        for (size_t i = 0; i < n; i++) {  // to_transcript_bytes of a field element: to_repr() reversed
            uint8_t be[32];
            const uint8_t* le = (const uint8_t*)scalars32_canonical + 32 * i;
            for (int k = 0; k < 32; k++) be[k] = le[31 - k];
            t->hasher.update(be, 32);
        }
    });
}
int lurk_hip_keccak_transcript_absorb_point(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, int curve, const void* point_jacobian96) {
    return host_guarded([&] {
        LURK_REQUIRE(t && (label || label_len == 0) && point_jacobian96, "null argument");
        uint8_t xy[64], buf[65];
        LURK_REQUIRE(lurk_hip_point_to_affine_canonical(curve, xy, point_jacobian96) == 0, lurk_hip_last_error());
        bool finite = false;
        for (int k = 0; k < 64; k++) finite |= xy[k] != 0;  // the identity comes back as (0, 0): to_coordinates' (0, 0, is_infinity)
        for (int k = 0; k < 32; k++) { buf[k] = xy[31 - k]; buf[32 + k] = xy[63 - k]; }
        buf[64] = finite ? 1 : 0;
        t->hasher.update(label, label_len);
        t->hasher.update(buf, 65);
    });
}
int lurk_hip_keccak_transcript_dom_sep(lurk_hip_keccak_transcript* t, const void* bytes, size_t len) {
    return host_guarded([&] {
        LURK_REQUIRE(t && (bytes || len == 0), "null argument");
        t->hasher.update("NoDS", 4);
        t->hasher.update(bytes, len);
    });
}
int lurk_hip_keccak_transcript_squeeze(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, int field_id, void* out32_canonical) {
    return host_guarded([&] {
        LURK_REQUIRE(t && (label || label_len == 0) && out32_canonical, "null argument");
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(t->round != 0xffff, "the transcript's round counter would overflow");
        std::vector<uint8_t> in = {'N', 'o', 'D', 'S', (uint8_t)(t->round & 0xff), (uint8_t)(t->round >> 8)};
        in.insert(in.end(), t->state, t->state + 64);
        in.insert(in.end(), (const uint8_t*)label, (const uint8_t*)label + label_len);
        uint8_t out[64];
        keccak_updated_state(t->hasher, in.data(), in.size(), out);
        t->round++;
        memcpy(t->state, out, 64);
        t->hasher = Keccak256();
        uint64_t r[4];
        if (field_id == 0) from_uniform64<PallasFp>(out, r);
        else if (field_id == 1) from_uniform64<PallasFq>(out, r);
        else from_uniform64<Bn254Fr>(out, r);
        memcpy(out32_canonical, r, 32);
    });
}

static int keccak_round_finish(lurk_hip_keccak_round_binding* b, void* out_r32_canonical) {
    const int rc = lurk_hip_keccak_transcript_squeeze(b->transcript, b->squeeze_label, b->squeeze_label_len, b->field_id, out_r32_canonical);
    if (rc != 0) return rc;
    if (b->challenges_out) {
        if (b->n_rounds >= b->challenges_cap) return LURK_HIP_ERR_INVALID_ARG;
        memcpy((char*)b->challenges_out + 32 * b->n_rounds, out_r32_canonical, 32);
    }
    b->n_rounds++;
    return 0;
}
int lurk_hip_keccak_sumcheck_challenge(void* binding, int round, const void* coefficients32_canonical, void* out_r32_canonical) {
    (void)round;
    auto* b = (lurk_hip_keccak_round_binding*)binding;
    if (!b || !b->transcript || !coefficients32_canonical || !out_r32_canonical || b->n_scalars < 1) return LURK_HIP_ERR_INVALID_ARG;
    const int rc = lurk_hip_keccak_transcript_absorb_scalars(b->transcript, b->absorb_label, b->absorb_label_len, coefficients32_canonical, (size_t)b->n_scalars);
    return rc != 0 ? rc : keccak_round_finish(b, out_r32_canonical);
}
int lurk_hip_keccak_ipa_challenge(void* binding, int round, const void* l_jacobian96, const void* r_jacobian96, void* out_r32_canonical) {
    (void)round;
    auto* b = (lurk_hip_keccak_round_binding*)binding;
    if (!b || !b->transcript || !l_jacobian96 || !r_jacobian96 || !out_r32_canonical) return LURK_HIP_ERR_INVALID_ARG;
    int rc = lurk_hip_keccak_transcript_absorb_point(b->transcript, b->absorb_label, b->absorb_label_len, b->curve, l_jacobian96);
    if (rc == 0) rc = lurk_hip_keccak_transcript_absorb_point(b->transcript, b->absorb_label2, b->absorb_label2_len, b->curve, r_jacobian96);
    return rc != 0 ? rc : keccak_round_finish(b, out_r32_canonical);
}
}

// spartan.hip - the single-instance compressing prover as host code of the library.
//
// What CompressedSNARK::prove runs on the primary curve (/root/reference/src/proof/nova.rs:341-356 -> arecibo RelaxedR1CSSNARK::prove):
// outer cubic sum-check of eq(tau) (Az Bz - (u Cz + E)), inner quadratic sum-check of (A + r B + r^2 C)(r_x, .) z, the two evaluation
// claims (W at r_y[1..], E at r_x) batched to one point by a third sum-check, one inner-product argument under the resident key.
// The PROTOCOL (labels, order of what is absorbed, zero padding of the two polynomials to a common length) is this repository's own,
// restated in oracle/spartan_ref.py / oracle/spartan_fast.py, whose prover this one equals element for element and whose verifier
// accepts its proofs; the transcript construction is arecibo's Keccak256Transcript (transcript.hip).  Parity unpinned upstream.
//
// Everything here is a sequence of the library's own entry points - multiply_vec on the shape and on its transpose, eq_evals, fold_vec,
// inner products, the sum-check and inner-product round loops with the Keccak round bindings - with the vectors resident and the
// field arithmetic of the claims on the host: lurk_beta_amd/spartan.py: SpartanProver.prove did the same from Python (0.4 ms of
// interpreter between the sum-checks, 0.8 ms between proofs).
#include <memory>
#include <vector>

#include "common.hpp"
#include "field.cuh"

namespace lurk {

static void sp_ok(int rc) {
    if (rc != 0) throw HipFailure{rc, lurk_hip_last_error()};
}

using SpScratch = ArenaBuf;  // scratch vectors come off the stream's arena (common.hpp): a push, not a hipMallocAsync

struct SpTranscript {
    lurk_hip_keccak_transcript* t = nullptr;
    ~SpTranscript() {
        if (t) (void)lurk_hip_keccak_transcript_destroy(t);
    }
};

template <class F>
struct SpField {
    static Fe<F> from_canonical(const void* p) {
        Fe<F> x;
        memcpy(x.l, p, 32);
        LURK_REQUIRE(!fe_canonical_ge_mod<F>(x.l), "a field element is not reduced modulo the field order");
        return fe_to_mont<F>(x);
    }
    static void to_canonical(const Fe<F>& m, void* out) {
        const Fe<F> c = fe_from_mont<F>(m);
        memcpy(out, c.l, 32);
    }
};

template <class F>
static Fe<F> sp_squeeze(lurk_hip_keccak_transcript* t, const char* label, int field_id) {
    uint64_t r[4];
    sp_ok(lurk_hip_keccak_transcript_squeeze(t, label, strlen(label), field_id, r));
    return SpField<F>::from_canonical(r);
}
template <class F>
static void sp_absorb(lurk_hip_keccak_transcript* t, const char* label, const std::vector<Fe<F>>& mont_vals) {
    std::vector<uint64_t> can(4 * mont_vals.size());
    for (size_t i = 0; i < mont_vals.size(); i++) SpField<F>::to_canonical(mont_vals[i], can.data() + 4 * i);
    sp_ok(lurk_hip_keccak_transcript_absorb_scalars(t, label, strlen(label), can.data(), mont_vals.size()));
}
// eq(point) as 2^|point| Montgomery elements on the device
template <class F>
static void sp_eq(int field_id, const std::vector<Fe<F>>& point, void* d_out, hipStream_t s) {
    // (the point is copied out of pageable memory before the call returns; every caller's vector outlives the proof anyway)
    sp_ok(lurk_hip_eq_evals_dev(field_id, point.empty() ? nullptr : (const void*)point.data(), (int)point.size(), d_out, (void*)s));
}
// the multilinear extension of a device table at `point` (Montgomery): <table, eq(point)>
template <class F>
static Fe<F> sp_mle(int field_id, const void* d_table, const std::vector<Fe<F>>& point, hipStream_t s) {
    SpScratch eq(((size_t)32) << point.size(), s);
    sp_eq<F>(field_id, point, eq.p, s);
    Fe<F> out;
    sp_ok(lurk_hip_inner_product_dev(field_id, d_table, eq.p, (size_t)1 << point.size(), out.l, (void*)s));
    return out;
}
static int sp_log2(size_t n) {
    int k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}

template <class F>
static void spartan_prove(int curve, int field_id, const lurk_hip_r1cs* shape, const lurk_hip_r1cs* shape_t, size_t nc, size_t nv, size_t nio,
                          lurk_hip_msm_ctx* key, const void* ck_c_jac96, const void* x_canonical, const void* u_canonical, const void* d_w,
                          const void* d_e, const void* comm_w_jac96, const void* comm_e_jac96, const void* label, size_t label_len,
                          lurk_hip_spartan_proof* out, hipStream_t s) {
    const int ell_x = sp_log2(nc), ell_y = sp_log2(nv) + 1;
    const size_t N = nc > nv ? nc : nv;
    const int ell = sp_log2(N);
    void* vs = (void*)s;
    stream_pool_retain();
    SpTranscript tr;
    sp_ok(lurk_hip_keccak_transcript_new(&tr.t, label, label_len));
    sp_ok(lurk_hip_keccak_transcript_absorb_point(tr.t, "comm_W", 6, curve, comm_w_jac96));
    sp_ok(lurk_hip_keccak_transcript_absorb_point(tr.t, "comm_E", 6, curve, comm_e_jac96));
    std::vector<Fe<F>> ux(1 + nio);
    ux[0] = SpField<F>::from_canonical(u_canonical);
    for (size_t i = 0; i < nio; i++) ux[1 + i] = SpField<F>::from_canonical((const char*)x_canonical + 32 * i);
    sp_absorb<F>(tr.t, "uX", ux);
    // z = [W | u | X | 0 ...] of length 2 num_vars
    SpScratch z(2 * nv * 32, s), az(nc * 32, s), bz(nc * 32, s), cz(nc * 32, s);
    LURK_HIP_CHECK(hipMemsetAsync(z.p, 0, 2 * nv * 32, s));
    LURK_HIP_CHECK(hipMemcpyAsync(z.p, d_w, nv * 32, hipMemcpyDeviceToDevice, s));
    LURK_HIP_CHECK(hipMemcpyAsync((char*)z.p + nv * 32, ux.data(), (1 + nio) * 32, hipMemcpyHostToDevice, s));
    sp_ok(lurk_hip_r1cs_multiply_vec_dev(shape, z.p, az.p, bz.p, cz.p, vs));
    std::vector<Fe<F>> tau(ell_x);
    for (int j = 0; j < ell_x; j++) tau[j] = sp_squeeze<F>(tr.t, "t", field_id);
    SpScratch d_tau(nc * 32, s), ucze(nc * 32, s);
    sp_eq<F>(field_id, tau, d_tau.p, s);
    sp_ok(lurk_hip_fold_vec_dev(field_id, d_e, cz.p, ux[0].l, nc, ucze.p, vs));  // E + u Cz
    // ---- outer sum-check: eq(tau) (Az Bz - (u Cz + E)), claim 0; Az and Bz are consumed, Cz and E are needed afterwards
    const uint64_t zero32[4] = {0, 0, 0, 0};
    std::vector<Fe<F>> r_x, r_y, r_z;
    auto challenges = [&](const std::vector<uint64_t>& buf, int rounds, std::vector<Fe<F>>& into) {
        into.resize(rounds);
        for (int j = 0; j < rounds; j++) into[j] = SpField<F>::from_canonical(buf.data() + 4 * j);
    };
    auto binding = [&](std::vector<uint64_t>& keep, int rounds, const char* absorb, const char* absorb2, const char* squeeze) {
        keep.assign((size_t)4 * (rounds > 0 ? rounds : 1), 0);
        lurk_hip_keccak_round_binding b;
        memset(&b, 0, sizeof(b));
        b.transcript = tr.t;
        b.field_id = field_id;
        b.curve = curve;
        b.absorb_label = absorb;
        b.absorb_label_len = strlen(absorb);
        b.absorb_label2 = absorb2;
        b.absorb_label2_len = absorb2 ? strlen(absorb2) : 0;
        b.squeeze_label = squeeze;
        b.squeeze_label_len = strlen(squeeze);
        b.challenges_out = keep.data();
        b.challenges_cap = (size_t)(rounds > 0 ? rounds : 1);
        return b;
    };
    uint64_t finals4[16], claim_out[4];
    {
        void* tabs[4] = {d_tau.p, az.p, bz.p, ucze.p};  // (consumed: nothing reads Az, Bz or the other two afterwards)
        std::vector<uint64_t> keep;
        lurk_hip_keccak_round_binding b = binding(keep, ell_x, "p", nullptr, "c");
        b.n_scalars = 4;
        sp_ok(lurk_hip_sumcheck_prove_dev(field_id, 3, tabs, nc, zero32, lurk_hip_keccak_sumcheck_challenge, &b, out->polys_outer, finals4, claim_out, vs));
        challenges(keep, ell_x, r_x);
    }
    const Fe<F> claim_az = SpField<F>::from_canonical(finals4 + 4), claim_bz = SpField<F>::from_canonical(finals4 + 8);
    const Fe<F> claim_cz = sp_mle<F>(field_id, cz.p, r_x, s), eval_e = sp_mle<F>(field_id, d_e, r_x, s);
    SpField<F>::to_canonical(claim_az, (char*)out->claims_outer);
    SpField<F>::to_canonical(claim_bz, (char*)out->claims_outer + 32);
    SpField<F>::to_canonical(claim_cz, (char*)out->claims_outer + 64);
    SpField<F>::to_canonical(eval_e, out->eval_e);
    sp_absorb<F>(tr.t, "claims_outer", {claim_az, claim_bz, claim_cz, eval_e});
    const Fe<F> r = sp_squeeze<F>(tr.t, "r", field_id), r2 = fe_mul<F>(r, r);
    const Fe<F> claim_inner = fe_add<F>(fe_add<F>(claim_az, fe_mul<F>(r, claim_bz)), fe_mul<F>(r2, claim_cz));
    // ---- inner sum-check: (A + r B + r^2 C)(r_x, .) z over the 2 num_vars columns: the transposed shape applied to eq(r_x)
    SpScratch abc(2 * nv * 32, s);
    {
        SpScratch eq_rx(nc * 32, s), ea(2 * nv * 32, s), eb(2 * nv * 32, s), ec(2 * nv * 32, s), ab(2 * nv * 32, s);
        sp_eq<F>(field_id, r_x, eq_rx.p, s);
        sp_ok(lurk_hip_r1cs_multiply_vec_dev(shape_t, eq_rx.p, ea.p, eb.p, ec.p, vs));
        sp_ok(lurk_hip_fold_vec_dev(field_id, ea.p, eb.p, r.l, 2 * nv, ab.p, vs));
        sp_ok(lurk_hip_fold_vec_dev(field_id, ab.p, ec.p, r2.l, 2 * nv, abc.p, vs));  // (the block's scratch is released in stream order)
    }
    {
        void* tabs[2] = {abc.p, z.p};  // (z is consumed: W itself is read below, not z)
        uint64_t claim_can[4];
        SpField<F>::to_canonical(claim_inner, claim_can);
        std::vector<uint64_t> keep;
        lurk_hip_keccak_round_binding b = binding(keep, ell_y, "p", nullptr, "c");
        b.n_scalars = 3;
        uint64_t fin2[8];
        sp_ok(lurk_hip_sumcheck_prove_dev(field_id, 2, tabs, 2 * nv, claim_can, lurk_hip_keccak_sumcheck_challenge, &b, out->polys_inner, fin2, claim_out, vs));
        challenges(keep, ell_y, r_y);
    }
    const std::vector<Fe<F>> r_y_tail(r_y.begin() + 1, r_y.end());
    const Fe<F> eval_w = sp_mle<F>(field_id, d_w, r_y_tail, s);
    SpField<F>::to_canonical(eval_w, out->eval_w);
    sp_absorb<F>(tr.t, "eval_W", {eval_w});
    // ---- the two evaluation claims -> one point: W and E zero-padded to N, their points padded with leading zeros
    SpScratch p1(N * 32, s), p2(N * 32, s);
    LURK_HIP_CHECK(hipMemsetAsync(p1.p, 0, N * 32, s));
    LURK_HIP_CHECK(hipMemsetAsync(p2.p, 0, N * 32, s));
    LURK_HIP_CHECK(hipMemcpyAsync(p1.p, d_w, nv * 32, hipMemcpyDeviceToDevice, s));
    LURK_HIP_CHECK(hipMemcpyAsync(p2.p, d_e, nc * 32, hipMemcpyDeviceToDevice, s));
    std::vector<Fe<F>> x1((size_t)ell - (ell_y - 1), fe_zero<F>()), x2((size_t)ell - ell_x, fe_zero<F>());
    x1.insert(x1.end(), r_y_tail.begin(), r_y_tail.end());
    x2.insert(x2.end(), r_x.begin(), r_x.end());
    const Fe<F> rho = sp_squeeze<F>(tr.t, "rho", field_id);
    uint64_t fin_batch[16];
    {
        SpScratch e1(N * 32, s), e2(N * 32, s), q1(N * 32, s), q2(N * 32, s);
        sp_eq<F>(field_id, x1, e1.p, s);
        sp_eq<F>(field_id, x2, e2.p, s);
        LURK_HIP_CHECK(hipMemcpyAsync(q1.p, p1.p, N * 32, hipMemcpyDeviceToDevice, s));
        LURK_HIP_CHECK(hipMemcpyAsync(q2.p, p2.p, N * 32, hipMemcpyDeviceToDevice, s));
        void* tabs[4] = {e1.p, q1.p, e2.p, q2.p};
        uint64_t coeffs[8], claim_can[4];
        const Fe<F> one = fe_one<F>();
        SpField<F>::to_canonical(one, coeffs);
        SpField<F>::to_canonical(rho, coeffs + 4);
        SpField<F>::to_canonical(fe_add<F>(eval_w, fe_mul<F>(rho, eval_e)), claim_can);
        std::vector<uint64_t> keep;
        lurk_hip_keccak_round_binding b = binding(keep, ell, "p", nullptr, "c");
        b.n_scalars = 3;
        sp_ok(lurk_hip_sumcheck_prove_batch_dev(field_id, 2, 2, tabs, N, coeffs, claim_can, lurk_hip_keccak_sumcheck_challenge, &b, out->polys_batch, fin_batch,
                                                claim_out, vs));
        challenges(keep, ell, r_z);
    }
    const Fe<F> ev1 = SpField<F>::from_canonical(fin_batch + 4), ev2 = SpField<F>::from_canonical(fin_batch + 12);  // (A_i(r), B_i(r)) per instance: the B's
    SpField<F>::to_canonical(ev1, (char*)out->evals_batch);
    SpField<F>::to_canonical(ev2, (char*)out->evals_batch + 32);
    sp_absorb<F>(tr.t, "evals_batch", {ev1, ev2});
    const Fe<F> gamma = sp_squeeze<F>(tr.t, "gamma", field_id);
    SpScratch joint(N * 32, s), eq_rz(N * 32, s);
    sp_ok(lurk_hip_fold_vec_dev(field_id, p1.p, p2.p, gamma.l, N, joint.p, vs));
    const Fe<F> r0 = sp_squeeze<F>(tr.t, "ipa_r0", field_id);
    uint64_t ck_c_scaled[12], ck_hat[8];
    sp_ok(lurk_hip_point_mul(curve, ck_c_scaled, ck_c_jac96, r0.l, 1));
    sp_eq<F>(field_id, r_z, eq_rz.p, s);
    std::vector<uint64_t> keep;
    lurk_hip_keccak_round_binding b = binding(keep, ell, "L", "R", "r");
    sp_ok(lurk_hip_ipa_prove_dev(key, joint.p, eq_rz.p, N, ck_c_scaled, lurk_hip_keccak_ipa_challenge, &b, out->ipa_l, out->ipa_r, out->ipa_a, ck_hat, vs));
    LURK_HIP_CHECK(hipStreamSynchronize(s));
}

// ---- the batched prover: several relaxed instances of different shapes and sizes, one key, ONE proof ---------------------------------
// The structure of arecibo's spartan::batched::BatchedRelaxedR1CSSNARK, which lurk-beta's SuperNova prover compresses with
// (/root/reference/src/proof/supernova.rs:110, 293-302): one outer (cubic) and one inner (quadratic) sum-check shared by all instances
// through powers of a challenge, every instance's two evaluation claims batched to one point, one opening under the resident key.
// = lurk_beta_amd/spartan.py: BatchedSpartanProver.prove call for call (which stays as the test mirror) = oracle/spartan_fast.py:
// prove_batched element for element; instances shorter than the largest are zero-padded, their points padded with leading zeros.
template <class F>
static void spartan_prove_batch(int curve, int field_id, const lurk_hip_spartan_instance* inst, size_t n, lurk_hip_msm_ctx* key, const void* ck_c_jac96,
                                const void* label, size_t label_len, lurk_hip_spartan_batch_proof* out, hipStream_t s) {
    void* vs = (void*)s;
    size_t max_nc = 0, max_nv = 0;
    for (size_t i = 0; i < n; i++) {
        max_nc = inst[i].num_cons > max_nc ? inst[i].num_cons : max_nc;
        max_nv = inst[i].num_vars > max_nv ? inst[i].num_vars : max_nv;
    }
    const int ell_x = sp_log2(max_nc), ell_y = sp_log2(max_nv) + 1;
    const size_t N = max_nc > max_nv ? max_nc : max_nv, LX = (size_t)1 << ell_x, LY = (size_t)1 << ell_y;
    const int ell = sp_log2(N);
    stream_pool_retain();
    SpTranscript tr;
    sp_ok(lurk_hip_keccak_transcript_new(&tr.t, label, label_len));
    {
        Fe<F> nn = fe_from_u64<F>((uint64_t)n);
        sp_absorb<F>(tr.t, "n", {nn});
    }
    std::vector<std::vector<Fe<F>>> ux(n);
    for (size_t i = 0; i < n; i++) {
        sp_ok(lurk_hip_keccak_transcript_absorb_point(tr.t, "comm_W", 6, curve, inst[i].comm_w_jacobian96));
        sp_ok(lurk_hip_keccak_transcript_absorb_point(tr.t, "comm_E", 6, curve, inst[i].comm_e_jacobian96));
        ux[i].resize(1 + inst[i].num_io);
        ux[i][0] = SpField<F>::from_canonical(inst[i].u32_canonical);
        for (size_t k = 0; k < inst[i].num_io; k++) ux[i][1 + k] = SpField<F>::from_canonical((const char*)inst[i].x32_canonical + 32 * k);
        sp_absorb<F>(tr.t, "uX", ux[i]);
    }
    std::vector<Fe<F>> tau(ell_x);
    for (int j = 0; j < ell_x; j++) tau[j] = sp_squeeze<F>(tr.t, "t", field_id);
    const Fe<F> rho_o = sp_squeeze<F>(tr.t, "rho_outer", field_id);
    auto powers = [](const Fe<F>& b, size_t count) {
        std::vector<Fe<F>> v(count);
        Fe<F> acc = fe_one<F>();
        for (size_t k = 0; k < count; k++) { v[k] = acc; acc = fe_mul<F>(acc, b); }
        return v;
    };
    auto canon = [](const std::vector<Fe<F>>& v) {
        std::vector<uint64_t> c(4 * v.size());
        for (size_t k = 0; k < v.size(); k++) SpField<F>::to_canonical(v[k], c.data() + 4 * k);
        return c;
    };
    auto challenges = [&](const std::vector<uint64_t>& buf, int rounds, std::vector<Fe<F>>& into) {
        into.resize(rounds);
        for (int j = 0; j < rounds; j++) into[j] = SpField<F>::from_canonical(buf.data() + 4 * j);
    };
    auto binding = [&](std::vector<uint64_t>& keep, int rounds, const char* absorb, const char* absorb2, const char* squeeze, int n_scalars) {
        keep.assign((size_t)4 * (rounds > 0 ? rounds : 1), 0);
        lurk_hip_keccak_round_binding b;
        memset(&b, 0, sizeof(b));
        b.transcript = tr.t;
        b.field_id = field_id;
        b.curve = curve;
        b.n_scalars = n_scalars;
        b.absorb_label = absorb;
        b.absorb_label_len = strlen(absorb);
        b.absorb_label2 = absorb2;
        b.absorb_label2_len = absorb2 ? strlen(absorb2) : 0;
        b.squeeze_label = squeeze;
        b.squeeze_label_len = strlen(squeeze);
        b.challenges_out = keep.data();
        b.challenges_cap = (size_t)(rounds > 0 ? rounds : 1);
        return b;
    };
    typedef std::unique_ptr<SpScratch> Buf;
    auto mk = [&](size_t elems) { return Buf(new SpScratch(elems * 32, s)); };
    auto padded_copy = [&](const void* src, size_t have, size_t want) {  // a fresh table of `want` elements: src, then zeros
        Buf b = mk(want);
        if (want > have) LURK_HIP_CHECK(hipMemsetAsync((char*)b->p + have * 32, 0, (want - have) * 32, s));
        LURK_HIP_CHECK(hipMemcpyAsync(b->p, src, have * 32, hipMemcpyDeviceToDevice, s));
        return b;
    };
    const uint64_t zero32[4] = {0, 0, 0, 0};
    uint64_t claim_out[4];
    Buf d_tau = mk(LX);
    sp_eq<F>(field_id, tau, d_tau->p, s);
    // ---- outer sum-check over all instances: sum_i rho_o^i eq(tau) (Az_i Bz_i - (u_i Cz_i + E_i)), claim 0
    std::vector<Buf> zs(n), czs(n);
    std::vector<Fe<F>> r_x, r_y, r_z;
    std::vector<uint64_t> fin_outer(n * 16);
    {
        std::vector<Buf> keepers;
        std::vector<void*> tabs;
        for (size_t i = 0; i < n; i++) {
            const size_t nc = inst[i].num_cons, nv = inst[i].num_vars, nio = inst[i].num_io;
            zs[i] = mk(2 * nv);
            LURK_HIP_CHECK(hipMemsetAsync(zs[i]->p, 0, 2 * nv * 32, s));
            LURK_HIP_CHECK(hipMemcpyAsync(zs[i]->p, inst[i].d_w32_mont, nv * 32, hipMemcpyDeviceToDevice, s));
            LURK_HIP_CHECK(hipMemcpyAsync((char*)zs[i]->p + nv * 32, ux[i].data(), (1 + nio) * 32, hipMemcpyHostToDevice, s));
            Buf az = mk(LX), bz = mk(LX), ucze = mk(LX), tau_i = mk(LX);
            czs[i] = mk(nc);
            if (LX > nc) {
                LURK_HIP_CHECK(hipMemsetAsync((char*)az->p + nc * 32, 0, (LX - nc) * 32, s));
                LURK_HIP_CHECK(hipMemsetAsync((char*)bz->p + nc * 32, 0, (LX - nc) * 32, s));
                LURK_HIP_CHECK(hipMemsetAsync((char*)ucze->p + nc * 32, 0, (LX - nc) * 32, s));
            }
            sp_ok(lurk_hip_r1cs_multiply_vec_dev(inst[i].shape, zs[i]->p, az->p, bz->p, czs[i]->p, vs));
            sp_ok(lurk_hip_fold_vec_dev(field_id, inst[i].d_e32_mont, czs[i]->p, ux[i][0].l, nc, ucze->p, vs));
            LURK_HIP_CHECK(hipMemcpyAsync(tau_i->p, d_tau->p, LX * 32, hipMemcpyDeviceToDevice, s));  // (every instance binds its own copy)
            for (Buf* b : {&tau_i, &az, &bz, &ucze}) {
                tabs.push_back((*b)->p);
                keepers.push_back(std::move(*b));
            }
        }
        const std::vector<uint64_t> coeffs = canon(powers(rho_o, n));
        std::vector<uint64_t> keep;
        lurk_hip_keccak_round_binding b = binding(keep, ell_x, "p", nullptr, "c", 4);
        sp_ok(lurk_hip_sumcheck_prove_batch_dev(field_id, 3, n, tabs.data(), LX, coeffs.data(), zero32, lurk_hip_keccak_sumcheck_challenge, &b, out->polys_outer,
                                                fin_outer.data(), claim_out, vs));
        challenges(keep, ell_x, r_x);
    }
    Buf eq_rx = mk(LX);
    sp_eq<F>(field_id, r_x, eq_rx->p, s);
    std::vector<Fe<F>> cl_a(n), cl_b(n), cl_c(n), ev_e(n), ev_w(n);
    {
        std::vector<Fe<F>> flat;
        for (size_t i = 0; i < n; i++) {
            const size_t nc = inst[i].num_cons;
            const int px = ell_x - sp_log2(nc);
            cl_a[i] = SpField<F>::from_canonical(fin_outer.data() + 16 * i + 4);
            cl_b[i] = SpField<F>::from_canonical(fin_outer.data() + 16 * i + 8);
            sp_ok(lurk_hip_inner_product_dev(field_id, czs[i]->p, eq_rx->p, nc, cl_c[i].l, vs));  // Cz_i (padded) at r_x: the leading entries of eq(r_x)
            ev_e[i] = sp_mle<F>(field_id, inst[i].d_e32_mont, std::vector<Fe<F>>(r_x.begin() + px, r_x.end()), s);
            SpField<F>::to_canonical(cl_a[i], (char*)out->claims_outer + 96 * i);
            SpField<F>::to_canonical(cl_b[i], (char*)out->claims_outer + 96 * i + 32);
            SpField<F>::to_canonical(cl_c[i], (char*)out->claims_outer + 96 * i + 64);
            SpField<F>::to_canonical(ev_e[i], (char*)out->evals_e + 32 * i);
            flat.push_back(cl_a[i]);
            flat.push_back(cl_b[i]);
            flat.push_back(cl_c[i]);
        }
        flat.insert(flat.end(), ev_e.begin(), ev_e.end());
        sp_absorb<F>(tr.t, "claims_outer", flat);
    }
    const Fe<F> r = sp_squeeze<F>(tr.t, "r", field_id), r2 = fe_mul<F>(r, r);
    const Fe<F> rho_i = sp_squeeze<F>(tr.t, "rho_inner", field_id);
    // ---- inner sum-check: sum_i rho_i^i (A_i + r B_i + r^2 C_i)(r_x, .) z_i over 2^ell_y columns
    {
        std::vector<Buf> keepers;
        std::vector<void*> tabs;
        const std::vector<Fe<F>> pw = powers(rho_i, n);
        Fe<F> claim = fe_zero<F>();
        for (size_t i = 0; i < n; i++) {
            const size_t nv = inst[i].num_vars;
            Buf abc = mk(LY), zp = padded_copy(zs[i]->p, 2 * nv, LY);
            {
                SpScratch ea(2 * nv * 32, s), eb(2 * nv * 32, s), ec(2 * nv * 32, s), ab(2 * nv * 32, s);
                sp_ok(lurk_hip_r1cs_multiply_vec_dev(inst[i].shape_t, eq_rx->p, ea.p, eb.p, ec.p, vs));  // eq(r_x)'s first num_cons entries
                sp_ok(lurk_hip_fold_vec_dev(field_id, ea.p, eb.p, r.l, 2 * nv, ab.p, vs));
                sp_ok(lurk_hip_fold_vec_dev(field_id, ab.p, ec.p, r2.l, 2 * nv, abc->p, vs));
            }
            if (LY > 2 * nv) LURK_HIP_CHECK(hipMemsetAsync((char*)abc->p + 2 * nv * 32, 0, (LY - 2 * nv) * 32, s));
            const Fe<F> ci = fe_add<F>(fe_add<F>(cl_a[i], fe_mul<F>(r, cl_b[i])), fe_mul<F>(r2, cl_c[i]));
            claim = fe_add<F>(claim, fe_mul<F>(pw[i], ci));
            tabs.push_back(abc->p);
            tabs.push_back(zp->p);
            keepers.push_back(std::move(abc));
            keepers.push_back(std::move(zp));
        }
        const std::vector<uint64_t> coeffs = canon(pw);
        uint64_t claim_can[4];
        SpField<F>::to_canonical(claim, claim_can);
        std::vector<uint64_t> keep, fin(n * 8);
        lurk_hip_keccak_round_binding b = binding(keep, ell_y, "p", nullptr, "c", 3);
        sp_ok(lurk_hip_sumcheck_prove_batch_dev(field_id, 2, n, tabs.data(), LY, coeffs.data(), claim_can, lurk_hip_keccak_sumcheck_challenge, &b, out->polys_inner,
                                                fin.data(), claim_out, vs));
        challenges(keep, ell_y, r_y);
    }
    for (size_t i = 0; i < n; i++) {
        const int py = ell_y - (sp_log2(inst[i].num_vars) + 1);
        ev_w[i] = sp_mle<F>(field_id, inst[i].d_w32_mont, std::vector<Fe<F>>(r_y.begin() + py + 1, r_y.end()), s);
        SpField<F>::to_canonical(ev_w[i], (char*)out->evals_w + 32 * i);
    }
    sp_absorb<F>(tr.t, "evals_W", ev_w);
    // ---- all 2 n evaluation claims -> one point
    std::vector<Buf> polys(2 * n);
    std::vector<std::vector<Fe<F>>> points(2 * n);
    std::vector<Fe<F>> claims(2 * n);
    for (size_t i = 0; i < n; i++) {
        const size_t nc = inst[i].num_cons, nv = inst[i].num_vars;
        const int py = ell_y - (sp_log2(nv) + 1), px = ell_x - sp_log2(nc);
        polys[2 * i] = padded_copy(inst[i].d_w32_mont, nv, N);
        polys[2 * i + 1] = padded_copy(inst[i].d_e32_mont, nc, N);
        points[2 * i].assign((size_t)ell - sp_log2(nv), fe_zero<F>());
        points[2 * i].insert(points[2 * i].end(), r_y.begin() + py + 1, r_y.end());
        points[2 * i + 1].assign((size_t)ell - sp_log2(nc), fe_zero<F>());
        points[2 * i + 1].insert(points[2 * i + 1].end(), r_x.begin() + px, r_x.end());
        claims[2 * i] = ev_w[i];
        claims[2 * i + 1] = ev_e[i];
    }
    const Fe<F> rho = sp_squeeze<F>(tr.t, "rho", field_id);
    std::vector<uint64_t> fin_batch(2 * n * 8);
    {
        std::vector<Buf> keepers;
        std::vector<void*> tabs;
        const std::vector<Fe<F>> pw = powers(rho, 2 * n);
        Fe<F> claim = fe_zero<F>();
        for (size_t k = 0; k < 2 * n; k++) {
            Buf e = mk(N), q = padded_copy(polys[k]->p, N, N);
            sp_eq<F>(field_id, points[k], e->p, s);
            claim = fe_add<F>(claim, fe_mul<F>(pw[k], claims[k]));
            tabs.push_back(e->p);
            tabs.push_back(q->p);
            keepers.push_back(std::move(e));
            keepers.push_back(std::move(q));
        }
        const std::vector<uint64_t> coeffs = canon(pw);
        uint64_t claim_can[4];
        SpField<F>::to_canonical(claim, claim_can);
        std::vector<uint64_t> keep;
        lurk_hip_keccak_round_binding b = binding(keep, ell, "p", nullptr, "c", 3);
        sp_ok(lurk_hip_sumcheck_prove_batch_dev(field_id, 2, 2 * n, tabs.data(), N, coeffs.data(), claim_can, lurk_hip_keccak_sumcheck_challenge, &b, out->polys_batch,
                                                fin_batch.data(), claim_out, vs));
        challenges(keep, ell, r_z);
    }
    std::vector<Fe<F>> evals_batch(2 * n);
    for (size_t k = 0; k < 2 * n; k++) {
        evals_batch[k] = SpField<F>::from_canonical(fin_batch.data() + 8 * k + 4);  // (eq_k(r_z), poly_k(r_z)): the polynomial's
        SpField<F>::to_canonical(evals_batch[k], (char*)out->evals_batch + 32 * k);
    }
    sp_absorb<F>(tr.t, "evals_batch", evals_batch);
    const Fe<F> gamma = sp_squeeze<F>(tr.t, "gamma", field_id);
    Buf joint = padded_copy(polys[0]->p, N, N), eq_rz = mk(N);
    {
        Fe<F> g = gamma;
        for (size_t k = 1; k < 2 * n; k++) {
            sp_ok(lurk_hip_fold_vec_dev(field_id, joint->p, polys[k]->p, g.l, N, joint->p, vs));  // joint += gamma^k poly_k (element-wise: in place)
            g = fe_mul<F>(g, gamma);
        }
    }
    const Fe<F> r0 = sp_squeeze<F>(tr.t, "ipa_r0", field_id);
    uint64_t ck_c_scaled[12], ck_hat[8];
    sp_ok(lurk_hip_point_mul(curve, ck_c_scaled, ck_c_jac96, r0.l, 1));
    sp_eq<F>(field_id, r_z, eq_rz->p, s);
    std::vector<uint64_t> keep;
    lurk_hip_keccak_round_binding b = binding(keep, ell, "L", "R", "r", 0);
    sp_ok(lurk_hip_ipa_prove_dev(key, joint->p, eq_rz->p, N, ck_c_scaled, lurk_hip_keccak_ipa_challenge, &b, out->ipa_l, out->ipa_r, out->ipa_a, ck_hat, vs));
    LURK_HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_spartan_prove_dev(const lurk_hip_r1cs* shape, const lurk_hip_r1cs* shape_t, size_t num_cons, size_t num_vars, size_t num_io, lurk_hip_msm_ctx* key,
                               const void* ck_c_jacobian96, const void* x32_canonical, const void* u32_canonical, const void* d_w, const void* d_e,
                               const void* comm_w_jacobian96, const void* comm_e_jacobian96, const void* label, size_t label_len, lurk_hip_spartan_proof* out,
                               void* stream) {
    return guarded([&] {
        LURK_REQUIRE(shape && shape_t && key && ck_c_jacobian96 && u32_canonical && d_w && d_e && comm_w_jacobian96 && comm_e_jacobian96 && out, "null argument");
        LURK_REQUIRE(num_io == 0 || x32_canonical, "null public IO");
        LURK_REQUIRE(label || label_len == 0, "null label");
        LURK_REQUIRE(num_cons >= 2 && (num_cons & (num_cons - 1)) == 0 && num_vars >= 2 && (num_vars & (num_vars - 1)) == 0, "num_cons and num_vars must be powers of two >= 2");
        LURK_REQUIRE(1 + num_io <= num_vars, "the public IO does not fit the second half of z");
        LURK_REQUIRE(out->polys_outer && out->claims_outer && out->eval_e && out->polys_inner && out->eval_w && out->polys_batch && out->evals_batch && out->ipa_l &&
                         out->ipa_r && out->ipa_a,
                     "null output buffer");
        int curve = 0, bits = 0, device = 0;
        size_t points = 0;
        if (lurk_hip_msm_ctx_info(key, &curve, &points, &bits, nullptr) != 0 || lurk_hip_msm_ctx_device(key, &device) != 0)
            throw HipFailure{LURK_HIP_ERR_INVALID_ARG, lurk_hip_last_error()};
        LURK_REQUIRE(points >= (num_cons > num_vars ? num_cons : num_vars), "the key has fewer points than the padded polynomials have elements");
        // every scratch vector below is sized from (num_cons, num_vars, num_io) while the mat-vecs write what the shapes say: the two must agree
        // (the transposed shape is 2 num_vars rows over the num_cons columns, stored as num_vars = num_cons - 1, num_io = 0), and the shapes'
        // field must be the scalar field of the key's curve
        {
            const int want_field = curve == LURK_CURVE_PALLAS ? LURK_FIELD_PALLAS_FQ : LURK_FIELD_PALLAS_FP;
            int f = -1, ft = -1;
            size_t c = 0, v = 0, io = 0, ct = 0, vt = 0, iot = 0;
            if (lurk_hip_r1cs_dims(shape, &f, &c, &v, &io) != 0 || lurk_hip_r1cs_dims(shape_t, &ft, &ct, &vt, &iot) != 0)
                throw HipFailure{LURK_HIP_ERR_INVALID_ARG, lurk_hip_last_error()};
            LURK_REQUIRE(c == num_cons && v == num_vars && io == num_io, "shape: its (num_cons, num_vars, num_io) differ from the arguments");
            LURK_REQUIRE(ct == 2 * num_vars && vt + 1 + iot == num_cons, "shape_t: not the transpose of the shape (2 num_vars rows over num_cons columns)");
            LURK_REQUIRE(f == want_field && ft == want_field, "the shapes are not over the scalar field of the key's curve");
        }
        DeviceGuard dg(device);
        if (curve == LURK_CURVE_PALLAS)
            spartan_prove<PallasFq>(curve, LURK_FIELD_PALLAS_FQ, shape, shape_t, num_cons, num_vars, num_io, key, ck_c_jacobian96, x32_canonical, u32_canonical, d_w, d_e,
                                    comm_w_jacobian96, comm_e_jacobian96, label, label_len, out, (hipStream_t)stream);
        else
            spartan_prove<PallasFp>(curve, LURK_FIELD_PALLAS_FP, shape, shape_t, num_cons, num_vars, num_io, key, ck_c_jacobian96, x32_canonical, u32_canonical, d_w, d_e,
                                    comm_w_jacobian96, comm_e_jacobian96, label, label_len, out, (hipStream_t)stream);
    });
}

int lurk_hip_spartan_prove_batch_dev(const lurk_hip_spartan_instance* instances, size_t n_instances, lurk_hip_msm_ctx* key, const void* ck_c_jacobian96,
                                     const void* label, size_t label_len, lurk_hip_spartan_batch_proof* out, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(instances && n_instances >= 1 && n_instances <= 64 && key && ck_c_jacobian96 && out, "null argument, or not 1..64 instances");
        LURK_REQUIRE(label || label_len == 0, "null label");
        LURK_REQUIRE(out->polys_outer && out->claims_outer && out->evals_e && out->polys_inner && out->evals_w && out->polys_batch && out->evals_batch && out->ipa_l &&
                         out->ipa_r && out->ipa_a,
                     "null output buffer");
        int curve = 0, bits = 0, device = 0;
        size_t points = 0;
        if (lurk_hip_msm_ctx_info(key, &curve, &points, &bits, nullptr) != 0 || lurk_hip_msm_ctx_device(key, &device) != 0)
            throw HipFailure{LURK_HIP_ERR_INVALID_ARG, lurk_hip_last_error()};
        const int want_field = curve == LURK_CURVE_PALLAS ? LURK_FIELD_PALLAS_FQ : LURK_FIELD_PALLAS_FP;
        for (size_t i = 0; i < n_instances; i++) {
            const lurk_hip_spartan_instance& it = instances[i];
            LURK_REQUIRE(it.shape && it.shape_t && it.u32_canonical && it.d_w32_mont && it.d_e32_mont && it.comm_w_jacobian96 && it.comm_e_jacobian96, "null instance field");
            LURK_REQUIRE(it.num_io == 0 || it.x32_canonical, "null public IO");
            LURK_REQUIRE(it.num_cons >= 2 && (it.num_cons & (it.num_cons - 1)) == 0 && it.num_vars >= 2 && (it.num_vars & (it.num_vars - 1)) == 0,
                         "num_cons and num_vars must be powers of two >= 2");
            LURK_REQUIRE(1 + it.num_io <= it.num_vars, "the public IO does not fit the second half of z");
            LURK_REQUIRE(points >= (it.num_cons > it.num_vars ? it.num_cons : it.num_vars), "the key has fewer points than the padded polynomials have elements");
            int f = -1, ft = -1;
            size_t c = 0, v = 0, io = 0, ct = 0, vt = 0, iot = 0;
            if (lurk_hip_r1cs_dims(it.shape, &f, &c, &v, &io) != 0 || lurk_hip_r1cs_dims(it.shape_t, &ft, &ct, &vt, &iot) != 0)
                throw HipFailure{LURK_HIP_ERR_INVALID_ARG, lurk_hip_last_error()};
            LURK_REQUIRE(c == it.num_cons && v == it.num_vars && io == it.num_io, "shape: its (num_cons, num_vars, num_io) differ from the instance's");
            LURK_REQUIRE(ct == 2 * it.num_vars && vt + 1 + iot == it.num_cons, "shape_t: not the transpose of the shape (2 num_vars rows over num_cons columns)");
            LURK_REQUIRE(f == want_field && ft == want_field, "the shapes are not over the scalar field of the key's curve");
        }
        DeviceGuard dg(device);
        if (curve == LURK_CURVE_PALLAS)
            spartan_prove_batch<PallasFq>(curve, LURK_FIELD_PALLAS_FQ, instances, n_instances, key, ck_c_jacobian96, label, label_len, out, (hipStream_t)stream);
        else
            spartan_prove_batch<PallasFp>(curve, LURK_FIELD_PALLAS_FP, instances, n_instances, key, ck_c_jacobian96, label, label_len, out, (hipStream_t)stream);
    });
}
}

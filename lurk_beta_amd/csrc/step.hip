// step.hip - one Nova folding step on one curve of the cycle, device-resident.
//
// Reference loop: RecursiveSNARK::prove_step as lurk-beta drives it (/root/reference/src/proof/nova.rs:282-295, SuperNova:
// /root/reference/src/proof/supernova.rs:231-244).  Per curve a step is arecibo's NIFS::prove (un-vendored `nova` dependency,
// /root/reference/Cargo.toml:128; published Nova construction):
//     comm_W2          = commit(ck, W2)                                      fresh instance of this step's circuit
//     T, comm_T        = commit_T: T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1 ;  commit(ck, T)
//     r                = RO(pp_digest, U1, U2, comm_T)                       Poseidon sponge over the other field: HOST side
//     W <- W1 + r W2,  E <- E1 + r T,  u <- u1 + r,  X <- X1 + r X2          (comm_W, comm_E fold on the host: two points)
// The transcript sits in the middle, so the entry point has two halves: `begin` (both commitments in flight on the key's
// async slots, the cross term between them; returns when the two 96-byte commitments are on the host) and `finish(r)`.
// The running pair (z1 = [W1 | u1 | X1], E1) never leaves HBM; a step's W2 may already be there (lurk_hip_slot_witness_dev).
// A prover holds two of these contexts: Pallas (primary: the Lurk step circuit) and Vesta (secondary: ~10^4 constraints).
#include <chrono>
#include <memory>
#include <vector>

#include "common.hpp"
#include "field.cuh"
#include "nifs_pre.hpp"

namespace lurk {

static void ok(int rc) {
    if (rc != 0) throw HipFailure{rc, lurk_hip_last_error()};
}

// acc <- acc + x (Montgomery elements in host memory; r * 1 in Montgomery form is r itself)
template <class F>
static void host_add_mont(uint64_t* acc, const void* x) {
    Fe<F> a, b;
    memcpy(a.l, acc, 32);
    memcpy(b.l, x, 32);
    a = fe_add<F>(a, b);
    memcpy(acc, a.l, 32);
}

template <class F>
static void mont_one(void* out32) {
    Fe<F> o = fe_one<F>();
    memcpy(out32, o.l, 32);
}

}  // namespace lurk

using namespace lurk;

struct lurk_hip_fold_ctx {
    int curve = 0, field_id = 0, device = 0;
    lurk_hip_r1cs* shape = nullptr;    // borrowed
    lurk_hip_msm_ctx* key = nullptr;   // borrowed
    lurk_hip_msm_multi* mkey = nullptr;  // borrowed: the key cut across several devices (lurk_hip_fold_ctx_create_multi) instead of `key`
    struct ShardBuf {  // per slice of mkey, on the slice's device: a buffer and a copy stream for W2's slice [0] and for T's [1]
        int device = 0;
        size_t first = 0, count = 0;
        DevBuf buf[2];
        hipStream_t copy_stream[2] = {nullptr, nullptr};
        ~ShardBuf() {
            for (int k = 0; k < 2; k++)
                if (copy_stream[k]) {
                    int prev = 0;
                    (void)hipGetDevice(&prev);
                    (void)hipSetDevice(device);
                    (void)hipStreamDestroy(copy_stream[k]);
                    (void)hipSetDevice(prev);
                }
        }
    };
    std::vector<std::unique_ptr<ShardBuf>> shard_bufs;
    hipEvent_t t_ev = nullptr;         // multi-device key: T is complete on the context's stream
    // Helper keys (lurk_hip_fold_ctx_add_helper): the SAME commitment key resident on other devices.  The commitment of an instance
    // staged ahead (lurk_hip_fold_step_prefetch) then runs on helper (k mod H) - the staged ranges are pushed peer-to-peer into a
    // buffer on the helper's device and committed under its key - while this context's device folds the open step; the late ranges
    // and T stay here.  This is the reference's producer thread (nova.rs:306-317) spread over GPUs: the one way IVC itself gains from
    // several devices, since the steps are sequential.
    struct Helper {
        lurk_hip_msm_ctx* key = nullptr;
        int device = 0;
        DevBuf buf[2];                     // the staged ranges of fresh-instance buffer b, on the helper's device
        hipStream_t stream[2] = {nullptr, nullptr};
        ~Helper() {
            int prev = 0;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(device);
            for (int k = 0; k < 2; k++)
                if (stream[k]) (void)hipStreamDestroy(stream[k]);
            (void)hipSetDevice(prev);
        }
    };
    // the transcript in stages (nifs_pre.hpp): with a pp_digest set, begin absorbs U1 (and runs the permutation that completes) while the
    // device works, absorbs U2 when comm_W2 has arrived, and lurk_hip_fold_step_challenge / lurk_hip_fold_step finish behind comm_T
    bool has_pp = false;
    uint8_t pp_digest[32] = {0};
    NifsPre* pre = nullptr;
    // The late ranges' own key: the key's points at the late positions as a separate (small-commitment form) context, built the first
    // time a begin brings late ranges and kept while their layout stays the same: their commitment is then one launch over ~10^4 scalars
    // instead of a pass of the bucket pipeline over a num_vars-long vector that is zero everywhere else.
    lurk_hip_msm_ctx* late_key = nullptr;
    bool late_key_refused = false;  // the device could not hold the late ranges' small-form table: they go through slot 3 from then on
    std::vector<std::pair<size_t, size_t>> late_layout;
    DevBuf late_vals;
    lurk_hip_fold_submit_hook_fn submit_hook = nullptr;  // lurk_hip_fold_ctx_set_submit_hook
    void* submit_hook_user = nullptr;
    std::vector<std::unique_ptr<Helper>> helpers;
    size_t helper_next = 0;
    int helper_of[2] = {-1, -1};       // which helper commits the instance staged in buffer b (-1: this context's own key)
    size_t num_cons = 0, num_vars = 0, num_io = 0, ncols = 0;
    DevBuf z[2], e[2], t[2];           // running pair ping-pongs between two buffers (cur = index of the live one); T of the open step = t[tcur]
    // round 6: A z1, B z1, C z1 of the running instance stay resident (they fold linearly: A (z1 + r z2) = A z1 + r A z2), so a step's
    // cross term gathers from z2 alone (fold.hip: r1cs_cross_term_cached_kernel) and leaves A z2, B z2, C z2 for finish(r) to fold in
    // The fold of the cache rides in the NEXT step's cross-term launch (abc_pending: the previous step's products abc2[abc_buf] and its
    // challenge abc_r), and the folds of z and E - which the cross term does not read - run on fold_stream: nothing of finish(r) stands
    // between the transcript and the next cross term.  T and the step's products ping-pong (tcur) so that the side stream's fold of E
    // may still read T while the next cross term writes the other buffer.
    DevBuf abc1[3], abc2[2][3];
    bool cached = false, abc_pending = false;
    int tcur = 0, abc_buf = 0;
    uint64_t abc_r[4] = {0}, u_host[4] = {0};  // the pending challenge; the running u (Montgomery) as the host folds it
    hipStream_t fold_stream = nullptr;
    hipEvent_t zfold_ev = nullptr, efold_ev[2] = {nullptr, nullptr};  // the last fold of (z, E); the fold that last read t[k]
    bool efold_valid[2] = {false, false};
    DevBuf z2[2], zstaged[2], zpatch;  // fresh instances [W2 | 1 | X2]: the open step's and the one staged ahead; the staged ranges alone
                                       // (what their commitment reads while late ranges are written into z2); the late ranges alone
    int cur = 0;
    bool begun = false;
    // the running INSTANCE U1 = (comm_W, comm_E, u, X) beside the running witness: host copies, folded in finish(r) exactly as
    // RelaxedR1CSInstance::fold does (comm_W1 + r comm_W2, comm_E1 + r comm_T, u1 + r, X1 + r X2); the transcript absorbs them
    uint64_t comm_w[12] = {0}, comm_e[12] = {0};       // Jacobian, identity (z = 0): RelaxedR1CSInstance::default
    std::vector<uint64_t> ux;                          // [u | X] Montgomery, 1 + num_io elements
    std::vector<uint64_t> open_x2;                     // X2 of the open step
    uint64_t open_cw[12] = {0}, open_ct[12] = {0};     // comm_W2, comm_T of the open step
    uint64_t owed_r[4] = {0};                          // finish(r) leaves the instance fold to fold_instance_settle (off the device's critical path)
    bool instance_owed = false;
    // fresh instances staged ahead of their step (lurk_hip_fold_step_prefetch): buffer b commits on the key's slot 2 b, T on slot 1
    int staged[2] = {0, 0}, n_staged = 0, next_buf = 0, open_buf = 0;
    bool folded_valid[2] = {false, false}, submitted[2] = {false, false}, partial[2] = {false, false};
    char* pin = nullptr;               // pinned staging for what arrives in pageable host memory (u2, X2, late ranges): an async copy
    size_t pin_cap = 0;                // from pageable memory makes the runtime wait, which held the cross term back by ~0.5 ms
    // one staging stream per fresh-instance buffer: staging instance k + 1 (which waits for the kernels that produce its witness) must not
    // hold back the small uploads begin(k) sends behind instance k's own staging (measured: begin(k) then waited for witness k + 1)
    hipStream_t stream = nullptr, stage_stream[2] = {nullptr, nullptr};
    hipEvent_t patch_ev = nullptr;     // the late-range buffer is shared by all steps: zeroed behind step k, written by step k + 1 on the other stream
    bool patch_ev_valid = false;
    hipEvent_t w2_ready = nullptr, staged_ev[2] = {nullptr, nullptr}, folded_ev[2] = {nullptr, nullptr};
    std::recursive_mutex mu;           // recursive: the submit hook runs inside begin and may stage the next instance (lurk_hip_fold_step_prefetch)
    bool in_hook = false;
    ~lurk_hip_fold_ctx() {
        if (pre) nifs_pre_free(pre);
        if (pin) (void)hipHostFree(pin);
        if (stream) (void)hipStreamDestroy(stream);
        if (fold_stream) (void)hipStreamDestroy(fold_stream);
        if (zfold_ev) (void)hipEventDestroy(zfold_ev);
        for (int k = 0; k < 2; k++)
            if (efold_ev[k]) (void)hipEventDestroy(efold_ev[k]);
        for (int k = 0; k < 2; k++)
            if (stage_stream[k]) (void)hipStreamDestroy(stage_stream[k]);
        if (patch_ev) (void)hipEventDestroy(patch_ev);
        if (t_ev) (void)hipEventDestroy(t_ev);
        if (w2_ready) (void)hipEventDestroy(w2_ready);
        for (int k = 0; k < 2; k++) {
            if (staged_ev[k]) (void)hipEventDestroy(staged_ev[k]);
            if (folded_ev[k]) (void)hipEventDestroy(folded_ev[k]);
        }
    }
};

namespace lurk {

// [u | X] <- [u1 + r | X1 + r X2] on the host (1 + num_io elements; u2 = 1)
template <class F>
static void fold_ux_host(std::vector<uint64_t>& ux, const std::vector<uint64_t>& x2, const void* r_mont) {
    Fe<F> r;
    memcpy(r.l, r_mont, 32);
    const size_t n = ux.size() / 4;
    for (size_t i = 0; i < n; i++) {
        Fe<F> a, b;
        memcpy(a.l, &ux[4 * i], 32);
        if (i == 0) b = fe_one<F>();
        else memcpy(b.l, &x2[4 * (i - 1)], 32);
        a = fe_add<F>(a, fe_mul<F>(r, b));
        memcpy(&ux[4 * i], a.l, 32);
    }
}

// U1 <- U1 + r U2 (RelaxedR1CSInstance::fold) for the step finish(r) closed last, if it is still owed
static void fold_instance_settle(lurk_hip_fold_ctx* c) {
    if (!c->instance_owed) return;
    // everything into temporaries first: if a group operation fails the fold stays owed and the instance is untouched (the device
    // witness was folded by finish(r) already, so a half-updated instance would silently diverge from it)
    uint64_t pair[24], new_w[12], new_e[12];
    memcpy(pair, c->comm_w, 96);
    ok(lurk_hip_point_mul(c->curve, pair + 12, c->open_cw, c->owed_r, 1));
    ok(lurk_hip_point_sum(c->curve, new_w, pair, 2));
    memcpy(pair, c->comm_e, 96);
    ok(lurk_hip_point_mul(c->curve, pair + 12, c->open_ct, c->owed_r, 1));
    ok(lurk_hip_point_sum(c->curve, new_e, pair, 2));
    std::vector<uint64_t> new_ux = c->ux;
    if (c->field_id == LURK_FIELD_PALLAS_FQ) fold_ux_host<PallasFq>(new_ux, c->open_x2, c->owed_r);
    else fold_ux_host<PallasFp>(new_ux, c->open_x2, c->owed_r);
    memcpy(c->comm_w, new_w, 96);
    memcpy(c->comm_e, new_e, 96);
    c->ux.swap(new_ux);
    c->instance_owed = false;
}

// r = RO(pp_digest, U1, U2, comm_T) in stages (nifs_pre.hpp).  begin: the settled running instance; a stale precomputation is dropped.
static void fold_challenge_drop(lurk_hip_fold_ctx* c) {
    if (c->pre) nifs_pre_free(c->pre);
    c->pre = nullptr;
}
static void fold_challenge_begin(lurk_hip_fold_ctx* c) {
    fold_challenge_drop(c);
    if (!c->has_pp) return;
    c->pre = nifs_pre_begin(c->curve, c->pp_digest, c->comm_w, c->comm_e, c->ux.data(), c->ux.data() + 4, c->num_io);
}
static void fold_challenge_finish(lurk_hip_fold_ctx* c, void* r32_mont) {
    LURK_REQUIRE(c->begun, "no step is open");
    if (c->pre) {
        nifs_pre_finish(c->pre, c->open_ct, r32_mont);
        fold_challenge_drop(c);
        return;
    }
    LURK_REQUIRE(c->has_pp, "no pp_digest: call lurk_hip_fold_ctx_set_pp_digest first");
    ok(lurk_hip_nifs_challenge(c->curve, c->pp_digest, c->comm_w, c->comm_e, c->ux.data(), c->ux.data() + 4, c->open_cw, c->open_x2.data(), c->num_io,
                               c->open_ct, r32_mont));
}

// the key and the slot the staged commitment of buffer b runs on
static lurk_hip_msm_ctx* fold_staged_key(lurk_hip_fold_ctx* c, int b) { return c->helper_of[b] >= 0 ? c->helpers[c->helper_of[b]]->key : c->key; }

// The scheduling class of a commitment staged AHEAD of its step (the next step's commit(W2) while a step is open).  Default
// LURK_MSM_SUBMIT_FOLLOW (round 6, msm.hip: submit_impl): its sort and plan run at once at the lowest wave priority, its accumulation -
// persistent, two waves per SIMD, lowest priority - starts when the open step's commit(T) has left the accumulate stage and fills the
// window the step's serial chain leaves (T's bucket reduction, the transcript, the folds, the next cross term), its tail keeps the raised
// priority (the next begin waits for it).  commit(T)'s accumulation keeps the VALU to itself.  Measured at rc = 100 with the witness
// producer two steps ahead (bench.py --stage-ahead 3): 3.03 ms per step against 3.24 with W2 committed inside its own step, 3.21 against
// 3.47 for both curves.  LURK_FOLD_STAGED_MODE=2: the round-2..5 behaviour (LURK_MSM_SUBMIT_BACKGROUND: persistent one-wave
// accumulation beside commit(T)'s: 3.9 ms).
static int fold_staged_mode() {
    static const int m = [] {
        const char* v = getenv("LURK_FOLD_STAGED_MODE");
        const int x = v ? atoi(v) : LURK_MSM_SUBMIT_FOLLOW;
        return x == LURK_MSM_SUBMIT_BACKGROUND || x == LURK_MSM_SUBMIT_DEFAULT || x == LURK_MSM_SUBMIT_FOREGROUND ? x : LURK_MSM_SUBMIT_FOLLOW;
    }();
    return m;
}

static void fold_submit_staged(lurk_hip_fold_ctx* c, int b, int mode) {
    if (c->submitted[b]) return;
    // (an instance staged whole is read in place: begin refuses late ranges for it, and what begin does write - u2 and X2 - lies
    // behind the num_vars elements a commitment, or a helper's peer copy, reads)
    const void* src = c->partial[b] ? c->zstaged[b].p : c->z2[b].p;
    if (c->helper_of[b] >= 0) {  // on another device: push the staged ranges there (ordered after their staging here), commit under the helper's key
        auto& h = *c->helpers[c->helper_of[b]];
        LURK_HIP_CHECK(hipEventRecord(c->staged_ev[b], c->stage_stream[b]));
        {
            DeviceGuard dg(h.device);
            LURK_HIP_CHECK(hipStreamWaitEvent(h.stream[b], c->staged_ev[b], 0));
            LURK_HIP_CHECK(hipMemcpyPeerAsync(h.buf[b].p, h.device, src, c->device, c->num_vars * 32, h.stream[b]));
        }
        ok(lurk_hip_msm_ctx_submit_dev_mode(h.key, 2 * b, h.buf[b].p, c->num_vars, 1, h.stream[b], LURK_MSM_SUBMIT_DEFAULT));
    } else {
        ok(lurk_hip_msm_ctx_submit_dev_mode(c->key, 2 * b, src, c->num_vars, 1, c->stage_stream[b], mode));  // zero digits cost the sort nothing
    }
    c->submitted[b] = true;
}

// Stage positions [offset, offset + count) of the next fresh witness (the rest zero for now) and start its commitment.
static void fold_stage(lurk_hip_fold_ctx* c, const void* w2, size_t offset, size_t count, int on_device, void* w2_stream, bool staged_ahead = false) {
    LURK_REQUIRE(c->n_staged < 2, "two fresh instances are already staged: begin a step first");
    // two z2 buffers: while a step is open its instance occupies one of them until finish(r) has folded it, so only ONE more
    // can be staged (the next buffer in turn would be the open step's own)
    LURK_REQUIRE(!((c->begun || c->in_hook) && c->n_staged >= 1), "a step is open and the other buffer is already staged: finish the step first");
    LURK_REQUIRE(offset <= c->num_vars && count <= c->num_vars - offset, "witness range out of bounds");
    LURK_REQUIRE(count == 0 || w2, "null witness");
    const int b = c->next_buf;
    char* z2 = (char*)c->z2[b].p;
    if (c->folded_valid[b]) LURK_HIP_CHECK(hipStreamWaitEvent(c->stage_stream[b], c->folded_ev[b], 0));  // the fold two steps back still reads it
    if (on_device) {  // W2 was produced on the caller's stream (e.g. by lurk_hip_slot_witness_dev): order ours after it
        LURK_HIP_CHECK(hipEventRecord(c->w2_ready, (hipStream_t)w2_stream));
        LURK_HIP_CHECK(hipStreamWaitEvent(c->stage_stream[b], c->w2_ready, 0));
    }
    if (count < c->num_vars) LURK_HIP_CHECK(hipMemsetAsync(z2, 0, c->num_vars * 32, c->stage_stream[b]));
    if (count)
        LURK_HIP_CHECK(hipMemcpyAsync(z2 + offset * 32, w2, count * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stage_stream[b]));
    c->partial[b] = count < c->num_vars;
    if (c->partial[b]) {  // late ranges will be written into z2 while the commitment of the staged ones is still in flight: it reads a copy
        if (!c->zstaged[b].p) c->zstaged[b].alloc(c->num_vars * 32);
        LURK_HIP_CHECK(hipMemcpyAsync(c->zstaged[b].p, z2, c->num_vars * 32, hipMemcpyDeviceToDevice, c->stage_stream[b]));
    }
    c->submitted[b] = false;
    c->helper_of[b] = -1;
    if (!c->helpers.empty() && staged_ahead) c->helper_of[b] = (int)(c->helper_next++ % c->helpers.size());
    c->staged[c->n_staged++] = b;
    c->next_buf = b ^ 1;
    if (c->helper_of[b] >= 0) {  // another device's queues: nothing this device waits for is delayed by starting now
        fold_submit_staged(c, b, LURK_MSM_SUBMIT_DEFAULT);
        return;
    }
    // Between begin and finish nothing the host waits for is in flight: the commitment starts now, in the background class.
    // Otherwise it is submitted by the begin that comes next, BEHIND that step's commit(T): the device serves its queues
    // roughly in submission order, and T is what the host waits for.
    if (c->begun || c->in_hook) fold_submit_staged(c, b, fold_staged_mode());
}

constexpr size_t FOLD_LATE_KEY_MAX = (size_t)1 << 16;  // late positions a key of their own is built for (the small-commitment form's limit)

// true: c->late_key commits exactly the positions of these patches, in their order
static bool fold_late_key(lurk_hip_fold_ctx* c, const lurk_hip_w2_patch* patches, size_t n_patches, size_t patched, hipStream_t s) {
    if (!c->key || patched == 0 || patched > FOLD_LATE_KEY_MAX || c->late_key_refused) return false;
    const char* sw = getenv("LURK_FOLD_LATE_KEY");  // 0: late ranges through the key's slot 3 as a num_vars-long vector (the form above 2^16 late positions)
    if (sw && atoi(sw) == 0) return false;
    std::vector<std::pair<size_t, size_t>> layout;
    for (size_t k = 0; k < n_patches; k++)
        if (patches[k].count) layout.emplace_back(patches[k].offset, patches[k].count);
    if (c->late_key && layout == c->late_layout) return true;
    if (c->late_key) {
        (void)lurk_hip_msm_ctx_destroy(c->late_key);
        c->late_key = nullptr;
    }
    const MsmTableView v = msm_ctx_table_view(c->key);  // the first npoints records of every form are the points themselves
    DevBuf pts(patched * 64);
    size_t at = 0;
    for (auto& r : layout) {
        LURK_REQUIRE(r.first + r.second <= v.npoints, "late range beyond the key");
        LURK_HIP_CHECK(hipMemcpyAsync((char*)pts.p + at * 64, (const char*)v.table + r.first * 64, r.second * 64, hipMemcpyDeviceToDevice, s));
        at += r.second;
    }
    LURK_HIP_CHECK(hipStreamSynchronize(s));
    // the small-commitment form, stated (not left to what happens to be free at this moment): a device that cannot hold its table
    // (256 KiB per late position) sends the late ranges through slot 3 of the key instead - the same answer on every later step
    if (lurk_hip_msm_ctx_create_dev(&c->late_key, c->curve, pts.p, patched, LURK_MSM_FLAG_PRECOMPUTE | LURK_MSM_FLAG_SMALL_FORM, (void*)s) != 0) {
        c->late_key = nullptr;
        if (last_error_code() != LURK_HIP_ERR_OOM) throw HipFailure{last_error_code(), lurk_hip_last_error()};
        (void)hipGetLastError();  // the refusal is handled here: nothing of it may resurface in a later launch check on this thread
        c->late_layout.clear();
        c->late_key_refused = true;
        return false;
    }
    c->late_layout = layout;
    return true;
}

// LURK_STEP_TRACE: absolute host times (us) of the step's seams on stderr
static void step_mark(const char* what) {
    static const bool trace = getenv("LURK_STEP_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "[mark] %.1f %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(), what);
}

// T of the open step on the context's stream: from the cached products of the running instance (z2's gathers alone; the previous
// step's fold of the cache rides in the same launch) or from (z1, z2)
static void fold_cross_term(lurk_hip_fold_ctx* c, const void* z2) {
    if (c->cached) {
        const int k = c->tcur;
        if (c->efold_valid[k]) LURK_HIP_CHECK(hipStreamWaitEvent(c->stream, c->efold_ev[k], 0));  // the fold of E two steps back read t[k]
        const DevBuf* prev = c->abc_pending ? c->abc2[c->abc_buf] : nullptr;
        ok(lurk_hip_r1cs_cross_term_cached_dev(c->shape, z2, c->abc1[0].p, c->abc1[1].p, c->abc1[2].p, c->u_host, prev ? prev[0].p : nullptr,
                                               prev ? prev[1].p : nullptr, prev ? prev[2].p : nullptr, prev ? c->abc_r : nullptr, c->t[k].p, c->abc2[k][0].p,
                                               c->abc2[k][1].p, c->abc2[k][2].p, c->stream));
        c->abc_pending = false;  // the launch is in the stream: a begin that fails later and is repeated must not fold the cache twice
    } else {
        ok(lurk_hip_r1cs_cross_term_dev(c->shape, c->z[c->cur].p, z2, c->t[c->tcur].p, c->stream));
    }
    LURK_HIP_CHECK(hipEventRecord(c->t_ev, c->stream));  // T (and the step's products) are complete
}

static void fold_run_submit_hook(lurk_hip_fold_ctx* c) {
    if (!c->submit_hook) return;
    struct InHook {
        lurk_hip_fold_ctx* c;
        ~InHook() { c->in_hook = false; }
    } guard{c};
    c->in_hook = true;  // lurk_hip_fold_step_prefetch from inside the hook stages AND submits: the open step's commitments are in flight
    LURK_REQUIRE(c->submit_hook(c->submit_hook_user) == 0, "the submit hook failed");
}

// The oldest staged instance becomes this step's: late ranges, u2 = 1, X2, then T and its commitment.
static void fold_begin(lurk_hip_fold_ctx* c, const lurk_hip_w2_patch* patches, size_t n_patches, const void* x2_mont, void* comm_w2_jac96,
                       void* comm_t_jac96) {
    LURK_REQUIRE(c->n_staged > 0, "no fresh instance is staged");
    static const bool trace = getenv("LURK_STEP_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[8] = {0};
    tt[0] = now();
    step_mark("begin");
    const int b = c->staged[0];
    char* z2 = (char*)c->z2[b].p;
    size_t patched = 0;
    for (size_t k = 0; k < n_patches; k++) {
        LURK_REQUIRE(patches[k].offset <= c->num_vars && patches[k].count <= c->num_vars - patches[k].offset, "patch out of bounds");
        LURK_REQUIRE(patches[k].count == 0 || patches[k].values, "null patch values");
        patched += patches[k].count;
    }
    LURK_REQUIRE(!patched || c->partial[b], "late ranges for an instance that was staged whole");
    uint64_t body[12], late[12];
    c->staged[0] = c->staged[1];
    c->n_staged--;
    // From here on the staged instance is consumed.  If anything below fails (a busy slot, an allocation, a device error) the
    // commitments this call has in flight are drained, the device work it enqueued is waited for, and the instance goes back to the
    // head of the staged queue with its commitment to be submitted again: the context, and the key's slots, stay usable and the
    // same begin can simply be repeated.
    struct Rollback {
        lurk_hip_fold_ctx* c;
        int b;
        const lurk_hip_w2_patch* patches;
        size_t n_patches;
        bool armed = true, t_in_flight = false, late_in_flight = false, patch_written = false, late_own_key = false;
        ~Rollback() {
            if (!armed) return;
            uint64_t junk[12];
            if (c->submitted[b]) (void)lurk_hip_msm_ctx_wait(fold_staged_key(c, b), 2 * b, junk);
            if (t_in_flight) (void)lurk_hip_msm_ctx_wait(c->key, 1, junk);
            if (late_in_flight) (void)(late_own_key ? lurk_hip_msm_ctx_wait(c->late_key, 0, junk) : lurk_hip_msm_ctx_wait(c->key, 3, junk));
            c->submitted[b] = false;
            if (patch_written && c->zpatch.p)
                for (size_t k = 0; k < n_patches; k++)
                    if (patches[k].count) (void)hipMemsetAsync((char*)c->zpatch.p + patches[k].offset * 32, 0, patches[k].count * 32, c->stage_stream[b]);
            (void)hipStreamSynchronize(c->stage_stream[b]);
            (void)hipStreamSynchronize(c->stream);
            c->staged[1] = c->staged[0];
            c->staged[0] = b;
            c->n_staged++;
            fold_challenge_drop(c);
        }
    } rollback{c, b, patches, n_patches};
    const bool ahead = c->n_staged > 0 && !c->submitted[c->staged[0]];  // the NEXT step's instance waits to be submitted behind T
    // u2 = 1 (a fresh instance is strict), X2 and the late ranges go through the pinned staging buffer
    const size_t need = (1 + c->num_io + patched) * 32;
    if (need > c->pin_cap) {
        if (c->pin) LURK_HIP_CHECK(hipHostFree(c->pin));
        c->pin = nullptr;
        c->pin_cap = 0;
        LURK_HIP_CHECK(hipHostMalloc((void**)&c->pin, need + need / 2, hipHostMallocDefault));
        c->pin_cap = need + need / 2;
    }
    if (c->field_id == LURK_FIELD_PALLAS_FQ) mont_one<PallasFq>(c->pin);
    else mont_one<PallasFp>(c->pin);
    if (c->num_io) memcpy(c->pin + 32, x2_mont, c->num_io * 32);
    LURK_HIP_CHECK(hipMemcpyAsync(z2 + c->num_vars * 32, c->pin, (1 + c->num_io) * 32, hipMemcpyHostToDevice, c->stage_stream[b]));
    const bool late_own_key = patched && fold_late_key(c, patches, n_patches, patched, c->stage_stream[b]);
    rollback.late_own_key = late_own_key;
    if (patched) {
        char* src = c->pin + (1 + c->num_io) * 32;
        if (late_own_key) {  // the late values, contiguous in patch order: what the late ranges' own key commits
            c->late_vals.ensure(patched * 32);
        } else {
            if (c->patch_ev_valid) LURK_HIP_CHECK(hipStreamWaitEvent(c->stage_stream[b], c->patch_ev, 0));  // the previous step's zeroing
            if (!c->zpatch.p) {
                c->zpatch.alloc(c->num_vars * 32);
                LURK_HIP_CHECK(hipMemsetAsync(c->zpatch.p, 0, c->num_vars * 32, c->stage_stream[b]));
            }
            rollback.patch_written = true;
        }
        for (size_t k = 0; k < n_patches; k++) {
            if (!patches[k].count) continue;
            memcpy(src, patches[k].values, patches[k].count * 32);
            LURK_HIP_CHECK(hipMemcpyAsync(z2 + patches[k].offset * 32, src, patches[k].count * 32, hipMemcpyHostToDevice, c->stage_stream[b]));
            if (!late_own_key)
                LURK_HIP_CHECK(hipMemcpyAsync((char*)c->zpatch.p + patches[k].offset * 32, src, patches[k].count * 32, hipMemcpyHostToDevice, c->stage_stream[b]));
            src += patches[k].count * 32;
        }
        if (late_own_key)
            LURK_HIP_CHECK(hipMemcpyAsync(c->late_vals.p, c->pin + (1 + c->num_io) * 32, patched * 32, hipMemcpyHostToDevice, c->stage_stream[b]));
    }
    LURK_HIP_CHECK(hipEventRecord(c->staged_ev[b], c->stage_stream[b]));
    LURK_HIP_CHECK(hipStreamWaitEvent(c->stream, c->staged_ev[b], 0));
    // Order (measured on MI355X at rc = 100, `profiles/r03_step_order_sweep.txt` and `profiles/r03e_step_experiments.txt`, ms per step with the next witness traced
    // behind begin): cross term first, then commit(W2), then commit(T), all in the FOREGROUND class: 3.88 - against commit(W2) first
    // 3.93; commit(W2) in the BACKGROUND class (persistent one-wave accumulation) 4.11 whichever comes first; commit(T) first and
    // commit(W2)'s accumulation held back until T's sort is through 4.15.  The cross term gathers from HBM while commit(W2) sorts;
    // its successor commit(T) is what the host waits for last.
    const int fg = LURK_MSM_SUBMIT_FOREGROUND;
    fold_cross_term(c, z2);                                                                     // T ...
    step_mark("cross term launched");
    fold_submit_staged(c, b, fg);
    tt[1] = now();
    ok(lurk_hip_msm_ctx_submit_dev_mode(c->key, 1, c->t[c->tcur].p, c->num_cons, 1, c->stream, fg));  // ... commit(T): what the host waits for
    rollback.t_in_flight = true;
    if (patched) {  // commitment of the late ranges: under their own key, or as a num_vars-long vector that is zero elsewhere
        if (late_own_key) ok(lurk_hip_msm_ctx_submit_dev_mode(c->late_key, 0, c->late_vals.p, patched, 1, c->stage_stream[b], fg));
        else ok(lurk_hip_msm_ctx_submit_dev_mode(c->key, 3, c->zpatch.p, c->num_vars, 1, c->stage_stream[b], fg));
        rollback.late_in_flight = true;
    }
    tt[2] = now();
    if (ahead) fold_submit_staged(c, c->staged[0], fold_staged_mode());  // commit(next W2) fills what T leaves
    tt[3] = now();
    fold_run_submit_hook(c);  // the caller's device work beside the step (the next witness's traces)
    fold_instance_settle(c);  // the previous step's instance fold, while the device works on this step
    fold_challenge_begin(c);  // ... and the part of this step's transcript that needs no commitment of this step
    if (patched) {
        c->submitted[b] = false;  // (a wait consumes the slot's commitment whether it succeeds or not)
        ok(lurk_hip_msm_ctx_wait(fold_staged_key(c, b), 2 * b, body));
        rollback.late_in_flight = false;
        ok(late_own_key ? lurk_hip_msm_ctx_wait(c->late_key, 0, late) : lurk_hip_msm_ctx_wait(c->key, 3, late));
        uint64_t two[24];
        memcpy(two, body, 96);
        memcpy(two + 12, late, 96);
        ok(lurk_hip_point_sum(c->curve, comm_w2_jac96, two, 2));  // commit is linear: body + late ranges
        if (!late_own_key) {
            for (size_t k = 0; k < n_patches; k++)
                if (patches[k].count) LURK_HIP_CHECK(hipMemsetAsync((char*)c->zpatch.p + patches[k].offset * 32, 0, patches[k].count * 32, c->stage_stream[b]));
            LURK_HIP_CHECK(hipEventRecord(c->patch_ev, c->stage_stream[b]));
            c->patch_ev_valid = true;
        }
    } else {
        c->submitted[b] = false;
        ok(lurk_hip_msm_ctx_wait(fold_staged_key(c, b), 2 * b, comm_w2_jac96));
    }
    if (c->pre) nifs_pre_fresh(c->pre, comm_w2_jac96, x2_mont);  // U2, while commit(T) is still running
    tt[4] = now();
    rollback.t_in_flight = false;
    ok(lurk_hip_msm_ctx_wait(c->key, 1, comm_t_jac96));
    rollback.armed = false;
    tt[5] = now();
    step_mark("T collected");
    if (trace)
        fprintf(stderr, "[step] copies %.0f us, cross+T submit %.0f, next-W2 submit %.0f, W2 wait %.0f, T wait %.0f\n", tt[1] - tt[0], tt[2] - tt[1],
                tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4]);
    c->open_buf = b;
    c->begun = true;
    c->open_x2.assign((const uint64_t*)x2_mont, (const uint64_t*)x2_mont + 4 * c->num_io);
    memcpy(c->open_cw, comm_w2_jac96, 96);
    memcpy(c->open_ct, comm_t_jac96, 96);
}

// begin = stage + open in one call: if opening fails, the instance this call staged is taken back as well (fold_begin's own
// rollback has returned it to the queue with nothing in flight), so that the caller can repeat the call as it stands
static void fold_stage_and_begin(lurk_hip_fold_ctx* c, const void* w2, int on_device, void* w2_stream, const void* x2_mont, void* comm_w2_jac96,
                                 void* comm_t_jac96) {
    const int n_before = c->n_staged, mine = c->next_buf;  // the buffer fold_stage is about to take
    fold_stage(c, w2, 0, c->num_vars, on_device, w2_stream);
    try {
        fold_begin(c, nullptr, 0, x2_mont, comm_w2_jac96, comm_t_jac96);
    } catch (...) {
        // fold_begin's rollback has put the instance it consumed back at the head of the queue with nothing of it in flight.  Take
        // back EXACTLY what this call added, whatever else is queued: the instance this call staged, and - when the submit hook
        // staged the next one before the failure (the queue was empty when the call started, so anything behind `mine` is the
        // hook's) - that one too, with its background commitment drained: the hook runs again when the call is repeated.
        auto drain = [&](int b) {
            if (c->submitted[b]) {
                uint64_t junk[12];
                (void)lurk_hip_msm_ctx_wait(fold_staged_key(c, b), 2 * b, junk);
                c->submitted[b] = false;
            }
            (void)hipStreamSynchronize(c->stage_stream[b]);
        };
        int kept[2], n_kept = 0;
        for (int k = 0; k < c->n_staged; k++) {
            const int b = c->staged[k];
            if (b == mine || n_before == 0) drain(b);
            else kept[n_kept++] = b;
        }
        for (int k = 0; k < n_kept; k++) c->staged[k] = kept[k];
        c->n_staged = n_kept;
        c->next_buf = mine;
        throw;
    }
}

// ---- the key cut across several devices ("witness-commitment batches shard across the GPUs") ----------------------
// The step's vectors live on the context's device (cross term, folds); a commitment pushes slice i of the vector peer-to-peer into
// slice i's device, the slices commit concurrently (one host thread per device inside lurk_hip_msm_multi) and the 96-byte partial
// commitments are summed with the host group law.  Nothing else crosses a link.
// Slice i of a vector of the step goes to slice i's device on that device's copy stream (ordered after `ready`, an event of the
// context's device) and its commitment is submitted behind the copy; nothing blocks: both commitments of a step are in flight on
// every device before the host waits for the first partial.
static void fold_submit_multi(lurk_hip_fold_ctx* c, int which, const void* d_vec, size_t n, hipEvent_t ready) {
    std::vector<const void*> ptrs(c->shard_bufs.size(), nullptr);
    std::vector<void*> streams(c->shard_bufs.size(), nullptr);
    for (size_t i = 0; i < c->shard_bufs.size(); i++) {
        auto& sb = *c->shard_bufs[i];
        ptrs[i] = sb.buf[which].p;
        streams[i] = (void*)sb.copy_stream[which];
        if (sb.first >= n || sb.count == 0) continue;
        const size_t cnt = (sb.first + sb.count < n ? sb.first + sb.count : n) - sb.first;
        DeviceGuard dg(sb.device);
        LURK_HIP_CHECK(hipStreamWaitEvent(sb.copy_stream[which], ready, 0));
        LURK_HIP_CHECK(hipMemcpyPeerAsync(sb.buf[which].p, sb.device, (const char*)d_vec + sb.first * 32, c->device, cnt * 32, sb.copy_stream[which]));
    }
    ok(lurk_hip_msm_multi_submit_dev(c->mkey, which, ptrs.data(), streams.data(), ptrs.size(), n, 1, LURK_MSM_SUBMIT_FOREGROUND));
}

static void fold_begin_multi(lurk_hip_fold_ctx* c, const void* w2, int on_device, void* w2_stream, const void* x2_mont, void* comm_w2_jac96,
                             void* comm_t_jac96) {
    const int b = 0;  // one fresh-instance buffer: nothing is staged ahead with a multi-device key
    char* z2 = (char*)c->z2[b].p;
    if (c->folded_valid[b]) LURK_HIP_CHECK(hipStreamWaitEvent(c->stage_stream[0], c->folded_ev[b], 0));  // the previous fold still reads it
    if (on_device) {
        LURK_HIP_CHECK(hipEventRecord(c->w2_ready, (hipStream_t)w2_stream));
        LURK_HIP_CHECK(hipStreamWaitEvent(c->stage_stream[0], c->w2_ready, 0));
    }
    if (c->num_vars) LURK_HIP_CHECK(hipMemcpyAsync(z2, w2, c->num_vars * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stage_stream[0]));
    const size_t need = (1 + c->num_io) * 32;
    if (need > c->pin_cap) {
        if (c->pin) LURK_HIP_CHECK(hipHostFree(c->pin));
        c->pin = nullptr;
        c->pin_cap = 0;
        LURK_HIP_CHECK(hipHostMalloc((void**)&c->pin, need * 2, hipHostMallocDefault));
        c->pin_cap = need * 2;
    }
    if (c->field_id == LURK_FIELD_PALLAS_FQ) mont_one<PallasFq>(c->pin);
    else mont_one<PallasFp>(c->pin);
    if (c->num_io) memcpy(c->pin + 32, x2_mont, c->num_io * 32);
    LURK_HIP_CHECK(hipMemcpyAsync(z2 + c->num_vars * 32, c->pin, need, hipMemcpyHostToDevice, c->stage_stream[0]));
    LURK_HIP_CHECK(hipEventRecord(c->staged_ev[b], c->stage_stream[0]));
    LURK_HIP_CHECK(hipStreamWaitEvent(c->stream, c->staged_ev[b], 0));
    fold_cross_term(c, z2);  // beside the slices' copies and commit(W2)'s sorts (records t_ev)
    bool w_in_flight = false, t_in_flight = false;
    try {
        fold_submit_multi(c, 0, z2, c->num_vars, c->staged_ev[b]);
        w_in_flight = true;
        fold_submit_multi(c, 1, c->t[c->tcur].p, c->num_cons, c->t_ev);
        t_in_flight = true;
        fold_run_submit_hook(c);
        fold_instance_settle(c);  // the previous step's instance fold, while the devices work on this step
        fold_challenge_begin(c);
        w_in_flight = false;
        ok(lurk_hip_msm_multi_wait(c->mkey, 0, comm_w2_jac96));
        if (c->pre) nifs_pre_fresh(c->pre, comm_w2_jac96, x2_mont);
        t_in_flight = false;
        ok(lurk_hip_msm_multi_wait(c->mkey, 1, comm_t_jac96));
    } catch (...) {  // nothing stays in flight: the context and the key remain usable, the same begin can be repeated
        uint64_t junk[12];
        if (w_in_flight) (void)lurk_hip_msm_multi_wait(c->mkey, 0, junk);
        if (t_in_flight) (void)lurk_hip_msm_multi_wait(c->mkey, 1, junk);
        (void)hipStreamSynchronize(c->stage_stream[0]);
        (void)hipStreamSynchronize(c->stream);
        fold_challenge_drop(c);
        throw;
    }
    c->open_buf = b;
    c->begun = true;
    c->open_x2.assign((const uint64_t*)x2_mont, (const uint64_t*)x2_mont + 4 * c->num_io);
    memcpy(c->open_cw, comm_w2_jac96, 96);
    memcpy(c->open_ct, comm_t_jac96, 96);
}

static void fold_finish(lurk_hip_fold_ctx* c, const void* r32_mont) {
    const int nx = c->cur ^ 1;
    // z = [W | u | X]: one pass folds the witness, u <- u1 + r * 1 and X <- X1 + r X2
    const void* a[2] = {c->z[c->cur].p, c->e[c->cur].p};
    const void* b[2] = {c->z2[c->open_buf].p, c->t[c->tcur].p};
    void* o[2] = {c->z[nx].p, c->e[nx].p};
    const size_t n[2] = {c->ncols, c->num_cons};
    if (c->cached) {
        // off the chain: the next cross term reads the cached products (folded inside its own launch) and u (a kernel argument), not z or E
        LURK_HIP_CHECK(hipStreamWaitEvent(c->fold_stream, c->t_ev, 0));  // T of this step is complete (the cross term ran on the context's stream)
        LURK_HIP_CHECK(hipStreamWaitEvent(c->fold_stream, c->staged_ev[c->open_buf], 0));  // ... and so is z2 (staged on its own stream)
        ok(lurk_hip_fold_vecs_dev(c->field_id, 2, a, b, n, o, r32_mont, c->fold_stream));
        LURK_HIP_CHECK(hipEventRecord(c->folded_ev[c->open_buf], c->fold_stream));
        LURK_HIP_CHECK(hipEventRecord(c->efold_ev[c->tcur], c->fold_stream));
        LURK_HIP_CHECK(hipEventRecord(c->zfold_ev, c->fold_stream));
        c->efold_valid[c->tcur] = true;
        memcpy(c->abc_r, r32_mont, 32);
        c->abc_buf = c->tcur;
        c->abc_pending = true;
        c->tcur ^= 1;
        if (c->field_id == LURK_FIELD_PALLAS_FQ) host_add_mont<PallasFq>(c->u_host, r32_mont);  // u <- u + r u2, u2 = 1 (a fresh instance is strict)
        else host_add_mont<PallasFp>(c->u_host, r32_mont);
    } else {
        ok(lurk_hip_fold_vecs_dev(c->field_id, 2, a, b, n, o, r32_mont, c->stream));
        LURK_HIP_CHECK(hipEventRecord(c->folded_ev[c->open_buf], c->stream));
    }
    c->folded_valid[c->open_buf] = true;
    c->cur = nx;
    c->begun = false;
    // the instance side (two 128-bit scalar multiples, two additions, u and X: ~0.1 ms of host work) is owed until somebody reads the
    // instance or the next step has its device work enqueued - the next cross term must not wait for it
    memcpy(c->owed_r, r32_mont, 32);
    c->instance_owed = true;
    fold_challenge_drop(c);
}

}  // namespace lurk

extern "C" {

static void fold_ctx_create(lurk_hip_fold_ctx** out, int curve, lurk_hip_r1cs* shape, lurk_hip_msm_ctx* key, lurk_hip_msm_multi* mkey) {
    LURK_REQUIRE(out && shape && (key || mkey), "null argument");
    *out = nullptr;
    LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
    auto c = std::make_unique<lurk_hip_fold_ctx>();
    c->curve = curve;
    int shape_field = 0;
    ok(lurk_hip_r1cs_dims(shape, &shape_field, &c->num_cons, &c->num_vars, &c->num_io));
    c->field_id = curve == LURK_CURVE_PALLAS ? LURK_FIELD_PALLAS_FQ : LURK_FIELD_PALLAS_FP;  // the curve's scalar field
    LURK_REQUIRE(shape_field == c->field_id, "the shape is not over this curve's scalar field");
    c->shape = shape;
    c->key = key;
    c->mkey = mkey;
    c->ncols = c->num_vars + 1 + c->num_io;
    // the context lives where its shape lives; a single-device key must be there too (every other handle records its device as well)
    int shape_device = 0;
    ok(lurk_hip_r1cs_device(shape, &shape_device));
    if (key) {
        int key_device = 0;
        ok(lurk_hip_msm_ctx_device(key, &key_device));
        LURK_REQUIRE(key_device == shape_device, "the R1CS shape and the commitment key live on different devices");
    }
    c->device = shape_device;
    DeviceGuard dg(c->device);
    if (mkey) {
        const int ns = lurk_hip_msm_multi_num_shards(mkey);
        size_t total = 0;
        for (int i = 0; i < ns; i++) {
            auto sb = std::make_unique<lurk_hip_fold_ctx::ShardBuf>();
            ok(lurk_hip_msm_multi_shard(mkey, i, &sb->device, &sb->first, &sb->count));
            total += sb->count;
            DeviceGuard sg(sb->device);
            for (int k = 0; k < 2; k++) {
                sb->buf[k].alloc(sb->count * 32);
                LURK_HIP_CHECK(hipStreamCreateWithFlags(&sb->copy_stream[k], hipStreamNonBlocking));
            }
            c->shard_bufs.push_back(std::move(sb));
        }
        LURK_REQUIRE(total >= c->num_vars && total >= c->num_cons, "the multi-device key is shorter than the step's vectors");
    }
    for (int k = 0; k < 2; k++) {
        c->z[k].alloc(c->ncols * 32);
        c->e[k].alloc(c->num_cons * 32);
    }
    c->t[0].alloc(c->num_cons * 32);
    {
        const char* sw = getenv("LURK_FOLD_CACHED_PRODUCTS");  // 0: the six-gather cross term of rounds 1-5 (A/B runs, the parity test of both forms)
        c->cached = !(sw && atoi(sw) == 0);
    }
    if (c->cached) {
        c->t[1].alloc(c->num_cons * 32);
        for (int k = 0; k < 3; k++) {
            c->abc1[k].alloc(c->num_cons * 32);
            c->abc2[0][k].alloc(c->num_cons * 32);
            c->abc2[1][k].alloc(c->num_cons * 32);
        }
    }
    c->ux.assign(4 * (1 + c->num_io), 0);
    {   // the transcript's width-25 Poseidon constants are generated on first use (~0.15 s): now, not inside the first step
        uint64_t one[4] = {1, 0, 0, 0}, out[4];
        ok(lurk_hip_nova_ro_squeeze(c->field_id == LURK_FIELD_PALLAS_FQ ? LURK_FIELD_PALLAS_FP : LURK_FIELD_PALLAS_FQ, one, 1, 128, out));
    }
    LURK_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (c->cached) {
        LURK_HIP_CHECK(hipStreamCreateWithFlags(&c->fold_stream, hipStreamNonBlocking));
        LURK_HIP_CHECK(hipEventCreateWithFlags(&c->zfold_ev, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) LURK_HIP_CHECK(hipEventCreateWithFlags(&c->efold_ev[k], hipEventDisableTiming));
    }
    for (int k = 0; k < 2; k++) LURK_HIP_CHECK(hipStreamCreateWithFlags(&c->stage_stream[k], hipStreamNonBlocking));
    LURK_HIP_CHECK(hipEventCreateWithFlags(&c->patch_ev, hipEventDisableTiming));
    LURK_HIP_CHECK(hipEventCreateWithFlags(&c->w2_ready, hipEventDisableTiming));
    LURK_HIP_CHECK(hipEventCreateWithFlags(&c->t_ev, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
        c->z2[k].alloc(c->ncols * 32);
        LURK_HIP_CHECK(hipEventCreateWithFlags(&c->staged_ev[k], hipEventDisableTiming));
        LURK_HIP_CHECK(hipEventCreateWithFlags(&c->folded_ev[k], hipEventDisableTiming));
    }
    // RelaxedR1CSWitness::default / RelaxedR1CSInstance::default: W = 0, E = 0, u = 0, X = 0
    LURK_HIP_CHECK(hipMemsetAsync(c->z[0].p, 0, c->ncols * 32, c->stream));
    LURK_HIP_CHECK(hipMemsetAsync(c->e[0].p, 0, c->num_cons * 32, c->stream));
    if (c->cached)
        for (int k = 0; k < 3; k++) LURK_HIP_CHECK(hipMemsetAsync(c->abc1[k].p, 0, c->num_cons * 32, c->stream));  // A 0 = B 0 = C 0 = 0
    LURK_HIP_CHECK(hipStreamSynchronize(c->stream));
    *out = c.release();
}

int lurk_hip_fold_ctx_create(lurk_hip_fold_ctx** out, int curve, lurk_hip_r1cs* shape, lurk_hip_msm_ctx* key) {
    return guarded([&] {
        LURK_REQUIRE(key, "null key");
        fold_ctx_create(out, curve, shape, key, nullptr);
    });
}
int lurk_hip_fold_ctx_create_multi(lurk_hip_fold_ctx** out, int curve, lurk_hip_r1cs* shape, lurk_hip_msm_multi* key) {
    return guarded([&] {
        LURK_REQUIRE(key, "null key");
        fold_ctx_create(out, curve, shape, nullptr, key);
    });
}

int lurk_hip_fold_ctx_add_helper(lurk_hip_fold_ctx* c, lurk_hip_msm_ctx* helper_key) {
    return guarded([&] {
        LURK_REQUIRE(c && helper_key, "null argument");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->mkey, "helper keys belong to a context over a single-device key");
        LURK_REQUIRE(!c->begun && c->n_staged == 0, "add helper keys before the first step");
        int curve = 0, device = 0;
        size_t npoints = 0;
        ok(lurk_hip_msm_ctx_info(helper_key, &curve, &npoints, nullptr, nullptr));
        ok(lurk_hip_msm_ctx_device(helper_key, &device));
        LURK_REQUIRE(curve == c->curve, "the helper key is over another curve");
        LURK_REQUIRE(npoints >= c->num_vars, "the helper key has fewer points than the witness has elements");
        auto h = std::make_unique<lurk_hip_fold_ctx::Helper>();
        h->key = helper_key;
        h->device = device;
        {
            DeviceGuard dg(device);
            for (int k = 0; k < 2; k++) {
                h->buf[k].alloc(c->num_vars * 32);
                LURK_HIP_CHECK(hipStreamCreateWithFlags(&h->stream[k], hipStreamNonBlocking));
            }
        }
        c->helpers.push_back(std::move(h));
    });
}

int lurk_hip_fold_ctx_destroy(lurk_hip_fold_ctx* c) {
    if (!c) return 0;
    return guarded([&] {
        DeviceGuard dg(c->device);
        if (c->late_key) (void)lurk_hip_msm_ctx_destroy(c->late_key);
        delete c;
    });
}

// running pair <- host values (Montgomery): z1 = [W1 | u1 | X1] (num_vars + 1 + num_io elements), E1 (num_cons elements)
int lurk_hip_fold_ctx_set_running(lurk_hip_fold_ctx* c, const void* z1, const void* e1) {
    return guarded([&] {
        LURK_REQUIRE(c && z1 && e1, "null argument");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun, "a step is open: finish it first");
        if (c->fold_stream) LURK_HIP_CHECK(hipStreamSynchronize(c->fold_stream));  // the last step's folds of z and E write the buffers replaced here
        LURK_HIP_CHECK(hipMemcpyAsync(c->z[c->cur].p, z1, c->ncols * 32, hipMemcpyHostToDevice, c->stream));
        LURK_HIP_CHECK(hipMemcpyAsync(c->e[c->cur].p, e1, c->num_cons * 32, hipMemcpyHostToDevice, c->stream));
        if (c->cached) {  // the cache of the new running instance, whole; whatever fold of the old one was pending is void
            ok(lurk_hip_r1cs_multiply_vec_dev(c->shape, c->z[c->cur].p, c->abc1[0].p, c->abc1[1].p, c->abc1[2].p, c->stream));
            c->abc_pending = false;
            memcpy(c->u_host, (const char*)z1 + c->num_vars * 32, 32);
        }
        LURK_HIP_CHECK(hipStreamSynchronize(c->stream));
        fold_instance_settle(c);  // the commitments of the last step first: only u and X are replaced here
        memcpy(c->ux.data(), (const char*)z1 + c->num_vars * 32, (1 + c->num_io) * 32);
    });
}

// the running instance's commitments (96-byte Jacobians); u and X come with set_running's z1
int lurk_hip_fold_ctx_set_instance(lurk_hip_fold_ctx* c, const void* comm_w_jac96, const void* comm_e_jac96) {
    return guarded([&] {
        LURK_REQUIRE(c && comm_w_jac96 && comm_e_jac96, "null argument");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun, "a step is open: finish it first");
        fold_instance_settle(c);  // u and X of the last step first: only the commitments are replaced
        memcpy(c->comm_w, comm_w_jac96, 96);
        memcpy(c->comm_e, comm_e_jac96, 96);
    });
}

int lurk_hip_fold_ctx_instance(lurk_hip_fold_ctx* c, void* comm_w_jac96, void* comm_e_jac96, void* u32_mont, void* x_mont) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        fold_instance_settle(c);
        if (comm_w_jac96) memcpy(comm_w_jac96, c->comm_w, 96);
        if (comm_e_jac96) memcpy(comm_e_jac96, c->comm_e, 96);
        if (u32_mont) memcpy(u32_mont, c->ux.data(), 32);
        if (x_mont && c->num_io) memcpy(x_mont, c->ux.data() + 4, c->num_io * 32);
    });
}

// NIFS::prove in one call: begin -> r = RO(pp_digest, U1, U2, comm_T) (transcript.hip) -> finish(r)
int lurk_hip_fold_step(lurk_hip_fold_ctx* c, const void* w2, int w2_on_device, void* w2_stream, const void* x2_mont, const void* pp_digest32,
                       void* comm_w2_jac96, void* comm_t_jac96, void* r32_mont) {
    return guarded([&] {
        LURK_REQUIRE(c && pp_digest32, "null argument");
        LURK_REQUIRE(c->num_vars == 0 || w2, "null witness");
        LURK_REQUIRE(c->num_io == 0 || x2_mont, "null public IO");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun && !c->in_hook, "a step is already open: finish it first");
        LURK_REQUIRE(c->n_staged == 0, "fresh instances are staged: use lurk_hip_fold_step_begin_prefetched");
        uint64_t cw[12], ct[12], r[4];
        // the digest belongs to THIS call: only lurk_hip_fold_ctx_set_pp_digest makes the staged transcript a property of the context
        // (a caller with its own transcript that once used this entry point must not pay for - or fail in - nifs_pre_begin afterwards)
        struct DigestScope {
            lurk_hip_fold_ctx* c;
            bool had;
            uint8_t saved[32];
            DigestScope(lurk_hip_fold_ctx* cc, const void* d) : c(cc), had(cc->has_pp) {
                memcpy(saved, c->pp_digest, 32);
                memcpy(c->pp_digest, d, 32);
                c->has_pp = true;
            }
            ~DigestScope() {
                memcpy(c->pp_digest, saved, 32);
                c->has_pp = had;
                if (!had) fold_challenge_drop(c);
            }
        } digest_scope(c, pp_digest32);
        if (c->mkey) {
            fold_begin_multi(c, w2, w2_on_device, w2_stream, x2_mont, cw, ct);
        } else {
            fold_stage_and_begin(c, w2, w2_on_device, w2_stream, x2_mont, cw, ct);
        }
        fold_challenge_finish(c, r);  // U1 and U2 were absorbed while the device worked: one permutation behind comm_T
        fold_finish(c, r);
        if (comm_w2_jac96) memcpy(comm_w2_jac96, cw, 96);
        if (comm_t_jac96) memcpy(comm_t_jac96, ct, 96);
        if (r32_mont) memcpy(r32_mont, r, 32);
    });
}

int lurk_hip_fold_step_begin(lurk_hip_fold_ctx* c, const void* w2, int w2_on_device, void* w2_stream, const void* x2_mont,
                             void* comm_w2_jac96, void* comm_t_jac96) {
    return guarded([&] {
        LURK_REQUIRE(c && comm_w2_jac96 && comm_t_jac96, "null argument");
        LURK_REQUIRE(c->num_vars == 0 || w2, "null witness");
        LURK_REQUIRE(c->num_io == 0 || x2_mont, "null public IO");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun && !c->in_hook, "a step is already open: finish it first");
        LURK_REQUIRE(c->n_staged == 0, "fresh instances are staged: use lurk_hip_fold_step_begin_prefetched");
        if (c->mkey) {
            fold_begin_multi(c, w2, w2_on_device, w2_stream, x2_mont, comm_w2_jac96, comm_t_jac96);
            return;
        }
        fold_stage_and_begin(c, w2, w2_on_device, w2_stream, x2_mont, comm_w2_jac96, comm_t_jac96);
    });
}

int lurk_hip_fold_step_prefetch(lurk_hip_fold_ctx* c, const void* w2_range, size_t offset, size_t count, int on_device, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->mkey, "staging ahead is not available with a key cut across devices: give the context helper keys (lurk_hip_fold_ctx_add_helper)");
        fold_stage(c, w2_range, offset, count, on_device, stream, /*staged_ahead=*/true);
    });
}

int lurk_hip_fold_step_begin_prefetched(lurk_hip_fold_ctx* c, const lurk_hip_w2_patch* patches, size_t n_patches, const void* x2_mont,
                                        void* comm_w2_jac96, void* comm_t_jac96) {
    return guarded([&] {
        LURK_REQUIRE(c && comm_w2_jac96 && comm_t_jac96, "null argument");
        LURK_REQUIRE(n_patches == 0 || patches, "null patches");
        LURK_REQUIRE(c->num_io == 0 || x2_mont, "null public IO");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun && !c->in_hook, "a step is already open: finish it first");
        fold_begin(c, patches, n_patches, x2_mont, comm_w2_jac96, comm_t_jac96);
    });
}

int lurk_hip_fold_ctx_set_submit_hook(lurk_hip_fold_ctx* c, lurk_hip_fold_submit_hook_fn hook, void* user) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        c->submit_hook = hook;
        c->submit_hook_user = user;
    });
}

int lurk_hip_fold_ctx_set_pp_digest(lurk_hip_fold_ctx* c, const void* pp_digest32) {
    return guarded([&] {
        LURK_REQUIRE(c && pp_digest32, "null argument");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        memcpy(c->pp_digest, pp_digest32, 32);
        c->has_pp = true;
        fold_challenge_drop(c);
    });
}

int lurk_hip_fold_step_challenge(lurk_hip_fold_ctx* c, void* r32_mont) {
    return guarded([&] {
        LURK_REQUIRE(c && r32_mont, "null argument");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        step_mark("challenge");
        fold_challenge_finish(c, r32_mont);
        step_mark("challenge done");
    });
}

int lurk_hip_fold_step_finish(lurk_hip_fold_ctx* c, const void* r32_mont) {
    return guarded([&] {
        LURK_REQUIRE(c && r32_mont, "null argument");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_REQUIRE(c->begun, "no step is open");
        step_mark("finish");
        fold_finish(c, r32_mont);
        step_mark("finish done");
    });
}

int lurk_hip_fold_ctx_running_dev(lurk_hip_fold_ctx* c, void** d_z, void** d_e, void** stream) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        if (d_z) *d_z = c->z[c->cur].p;
        if (d_e) *d_e = c->e[c->cur].p;
        if (c->fold_stream) {  // the folds of z and E run on a side stream: the stream handed out is ordered behind the last of them
            DeviceGuard dg(c->device);
            LURK_HIP_CHECK(hipStreamWaitEvent(c->stream, c->zfold_ev, 0));
        }
        if (stream) *stream = (void*)c->stream;
    });
}

int lurk_hip_fold_ctx_read(lurk_hip_fold_ctx* c, void* z_host, void* e_host) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        DeviceGuard dg(c->device);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        LURK_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->fold_stream) LURK_HIP_CHECK(hipStreamSynchronize(c->fold_stream));
        if (z_host) LURK_HIP_CHECK(hipMemcpy(z_host, c->z[c->cur].p, c->ncols * 32, hipMemcpyDeviceToHost));
        if (e_host) LURK_HIP_CHECK(hipMemcpy(e_host, c->e[c->cur].p, c->num_cons * 32, hipMemcpyDeviceToHost));
    });
}
}

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
#include <memory>

#include "common.hpp"
#include "field.cuh"

namespace lurk {

static void ok(int rc) {
    if (rc != 0) throw HipFailure{rc, lurk_hip_last_error()};
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
    size_t num_cons = 0, num_vars = 0, num_io = 0, ncols = 0;
    DevBuf z[2], e[2], z2, t;          // running pair ping-pongs between two buffers (cur = index of the live one)
    int cur = 0;
    bool begun = false;
    hipStream_t stream = nullptr;
    hipEvent_t w2_ready = nullptr;
    std::mutex mu;
    ~lurk_hip_fold_ctx() {
        if (stream) (void)hipStreamDestroy(stream);
        if (w2_ready) (void)hipEventDestroy(w2_ready);
    }
};

extern "C" {

int lurk_hip_fold_ctx_create(lurk_hip_fold_ctx** out, int curve, lurk_hip_r1cs* shape, lurk_hip_msm_ctx* key) {
    return guarded([&] {
        LURK_REQUIRE(out && shape && key, "null argument");
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
        c->ncols = c->num_vars + 1 + c->num_io;
        c->device = current_device();
        for (int k = 0; k < 2; k++) {
            c->z[k].alloc(c->ncols * 32);
            c->e[k].alloc(c->num_cons * 32);
        }
        c->z2.alloc(c->ncols * 32);
        c->t.alloc(c->num_cons * 32);
        LURK_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        LURK_HIP_CHECK(hipEventCreateWithFlags(&c->w2_ready, hipEventDisableTiming));
        // RelaxedR1CSWitness::default / RelaxedR1CSInstance::default: W = 0, E = 0, u = 0, X = 0
        LURK_HIP_CHECK(hipMemsetAsync(c->z[0].p, 0, c->ncols * 32, c->stream));
        LURK_HIP_CHECK(hipMemsetAsync(c->e[0].p, 0, c->num_cons * 32, c->stream));
        LURK_HIP_CHECK(hipStreamSynchronize(c->stream));
        *out = c.release();
    });
}

int lurk_hip_fold_ctx_destroy(lurk_hip_fold_ctx* c) {
    if (!c) return 0;
    return guarded([&] {
        DeviceGuard dg(c->device);
        delete c;
    });
}

// running pair <- host values (Montgomery): z1 = [W1 | u1 | X1] (num_vars + 1 + num_io elements), E1 (num_cons elements)
int lurk_hip_fold_ctx_set_running(lurk_hip_fold_ctx* c, const void* z1, const void* e1) {
    return guarded([&] {
        LURK_REQUIRE(c && z1 && e1, "null argument");
        DeviceGuard dg(c->device);
        std::lock_guard<std::mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun, "a step is open: finish it first");
        LURK_HIP_CHECK(hipMemcpyAsync(c->z[c->cur].p, z1, c->ncols * 32, hipMemcpyHostToDevice, c->stream));
        LURK_HIP_CHECK(hipMemcpyAsync(c->e[c->cur].p, e1, c->num_cons * 32, hipMemcpyHostToDevice, c->stream));
        LURK_HIP_CHECK(hipStreamSynchronize(c->stream));
    });
}

int lurk_hip_fold_step_begin(lurk_hip_fold_ctx* c, const void* w2, int w2_on_device, void* w2_stream, const void* x2_mont,
                             void* comm_w2_jac96, void* comm_t_jac96) {
    return guarded([&] {
        LURK_REQUIRE(c && comm_w2_jac96 && comm_t_jac96, "null argument");
        LURK_REQUIRE(c->num_vars == 0 || w2, "null witness");
        LURK_REQUIRE(c->num_io == 0 || x2_mont, "null public IO");
        DeviceGuard dg(c->device);
        std::lock_guard<std::mutex> lk(c->mu);
        LURK_REQUIRE(!c->begun, "a step is already open: finish it first");
        char* z2 = (char*)c->z2.p;
        if (w2_on_device) {  // W2 was produced on the caller's stream (e.g. by lurk_hip_slot_witness_dev): order ours after it
            LURK_HIP_CHECK(hipEventRecord(c->w2_ready, (hipStream_t)w2_stream));
            LURK_HIP_CHECK(hipStreamWaitEvent(c->stream, c->w2_ready, 0));
        }
        if (c->num_vars)
            LURK_HIP_CHECK(hipMemcpyAsync(z2, w2, c->num_vars * 32, w2_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
        uint64_t one[4];
        if (c->field_id == LURK_FIELD_PALLAS_FQ) mont_one<PallasFq>(one);
        else mont_one<PallasFp>(one);
        LURK_HIP_CHECK(hipMemcpyAsync(z2 + c->num_vars * 32, one, 32, hipMemcpyHostToDevice, c->stream));  // u2 = 1 (a fresh instance is strict)
        if (c->num_io) LURK_HIP_CHECK(hipMemcpyAsync(z2 + (c->num_vars + 1) * 32, x2_mont, c->num_io * 32, hipMemcpyHostToDevice, c->stream));
        ok(lurk_hip_msm_ctx_submit_dev(c->key, 0, z2, c->num_vars, 1, c->stream));                            // commit(W2) ...
        ok(lurk_hip_r1cs_cross_term_dev(c->shape, c->z[c->cur].p, z2, c->t.p, c->stream));                    // ... T beside it ...
        ok(lurk_hip_msm_ctx_submit_dev(c->key, 1, c->t.p, c->num_cons, 1, c->stream));                       // ... commit(T)
        ok(lurk_hip_msm_ctx_wait(c->key, 0, comm_w2_jac96));
        ok(lurk_hip_msm_ctx_wait(c->key, 1, comm_t_jac96));
        c->begun = true;
    });
}

int lurk_hip_fold_step_finish(lurk_hip_fold_ctx* c, const void* r32_mont) {
    return guarded([&] {
        LURK_REQUIRE(c && r32_mont, "null argument");
        DeviceGuard dg(c->device);
        std::lock_guard<std::mutex> lk(c->mu);
        LURK_REQUIRE(c->begun, "no step is open");
        const int nx = c->cur ^ 1;
        // z = [W | u | X]: one pass folds the witness, u <- u1 + r * 1 and X <- X1 + r X2
        ok(lurk_hip_fold_vec_dev(c->field_id, c->z[c->cur].p, c->z2.p, r32_mont, c->ncols, c->z[nx].p, c->stream));
        ok(lurk_hip_fold_vec_dev(c->field_id, c->e[c->cur].p, c->t.p, r32_mont, c->num_cons, c->e[nx].p, c->stream));
        c->cur = nx;
        c->begun = false;
    });
}

int lurk_hip_fold_ctx_running_dev(lurk_hip_fold_ctx* c, void** d_z, void** d_e, void** stream) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        std::lock_guard<std::mutex> lk(c->mu);
        if (d_z) *d_z = c->z[c->cur].p;
        if (d_e) *d_e = c->e[c->cur].p;
        if (stream) *stream = (void*)c->stream;
    });
}

int lurk_hip_fold_ctx_read(lurk_hip_fold_ctx* c, void* z_host, void* e_host) {
    return guarded([&] {
        LURK_REQUIRE(c, "null ctx");
        DeviceGuard dg(c->device);
        std::lock_guard<std::mutex> lk(c->mu);
        LURK_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (z_host) LURK_HIP_CHECK(hipMemcpy(z_host, c->z[c->cur].p, c->ncols * 32, hipMemcpyDeviceToHost));
        if (e_host) LURK_HIP_CHECK(hipMemcpy(e_host, c->e[c->cur].p, c->num_cons * 32, hipMemcpyDeviceToHost));
    });
}
}

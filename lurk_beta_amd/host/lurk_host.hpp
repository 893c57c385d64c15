// lurk_host.hpp - C++ host-side mirror of the reference's interfaces for the hot path, written
// against the C ABI only (include/lurk_hip.h): what a Rust maintainer binds through FFI
//  expressed in the language available in this image.
//
//   lurk::host::PoseidonCache   <- PoseidonCache<F>::hash3/4/6/8, compute_hash  (/root/reference/src/hash.rs:86-204)
//   lurk::host::Trie            <- coprocessor::trie::Trie<F, 8, HEIGHT>        (/root/reference/src/coprocessor/trie/mod.rs:328-800)
//   lurk::host::CommitmentKey   <- arecibo CommitmentKey + CE::commit(ck, v)     (callers /root/reference/src/proof/nova.rs:287-293)
//   lurk::host::R1CSShape       <- arecibo R1CSShape::multiply_vec / commit_T cross term / witness fold (nova.rs:291-293)
//
// Field elements are 32-byte canonical little-endian values (Fe); points use the repr-c layouts.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lurk_hip.h"

namespace lurk {
namespace host {

struct Fe {
    std::array<uint64_t, 4> l{};
    Fe() = default;
    explicit Fe(uint64_t v) { l[0] = v; }
    bool operator<(const Fe& o) const { return l < o.l; }
    bool operator==(const Fe& o) const { return l == o.l; }
    bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
    bool bit(int i) const { return (l[i >> 6] >> (i & 63)) & 1; }
};

inline void check(int rc) {
    if (rc != 0) throw std::runtime_error(lurk_hip_last_error());  // the Rust side panics on a non-zero code
}

class PoseidonCache {
  public:
    explicit PoseidonCache(int field_id) : field_(field_id) {}
    Fe hash3(const std::array<Fe, 3>& p) { return compute(p.data(), 3); }
    Fe hash4(const std::array<Fe, 4>& p) { return compute(p.data(), 4); }
    Fe hash6(const std::array<Fe, 6>& p) { return compute(p.data(), 6); }
    Fe hash8(const std::array<Fe, 8>& p) { return compute(p.data(), 8); }
    // hash.rs:97-113: dispatch on the arity, anything but 3,4,6,8 is unreachable!/panic
    Fe compute_hash(const std::vector<Fe>& p) {
        if (p.size() != 3 && p.size() != 4 && p.size() != 6 && p.size() != 8) throw std::invalid_argument("unsupported arity");
        return compute(p.data(), (int)p.size());
    }
    // batched entry (store hydration hashes one DAG level per call)
    std::vector<Fe> hash_many(int arity, const std::vector<Fe>& flat_preimages) {
        size_t n = flat_preimages.size() / arity;
        std::vector<Fe> out(n);
        check(lurk_hip_poseidon_batch(field_, arity, flat_preimages.data(), n, out.data()));
        return out;
    }

  private:
    Fe compute(const Fe* p, int arity) {
        std::vector<Fe> key(p, p + arity);
        auto it = memo_.find(key);
        if (it != memo_.end()) return it->second;
        Fe d;
        check(lurk_hip_poseidon_batch(field_, arity, p, 1, &d));
        memo_.emplace(std::move(key), d);
        return d;
    }
    int field_;
    std::map<std::vector<Fe>, Fe> memo_;
};

// Sparse arity-8 Poseidon trie (StandardTrie = height 85).
class Trie {
  public:
    Trie(int field_id, int height, PoseidonCache& cache) : field_(field_id), height_(height), cache_(cache) {
        Fe cur;  // empty element = 0 (trie/mod.rs:430-433)
        for (int i = 0; i < height; i++) {  // init_empty (:464-481)
            std::array<Fe, 8> pre;
            pre.fill(cur);
            cur = reg(pre);
            empty_roots_.push_back(cur);
        }
        root_ = height ? empty_roots_.back() : Fe();
    }
    Fe root() const { return root_; }
    Fe empty_root_for_height(int h) const { return h == 0 ? Fe() : empty_roots_[h - 1]; }
    // path (:589-608): MSB-first bits, keep the last 3*H bits, 3-bit big-endian digits
    std::vector<int> path(const Fe& key) const {
        int nbits = field_ == LURK_FIELD_BN254_FR ? 254 : 255, need = 3 * height_;
        std::vector<int> be;
        for (int i = need - 1; i >= 0; i--) be.push_back(i < nbits ? key.bit(i) : 0);
        std::vector<int> out;
        for (int i = 0; i < need; i += 3) out.push_back(be[i] << 2 | be[i + 1] << 1 | be[i + 2]);
        return out;
    }
    bool lookup(const Fe& key, Fe* value) const {  // (:635-652)
        auto p = path(key);
        auto pres = along(p);
        *value = pres.back()[p.back()];
        return !value->is_zero();
    }
    bool insert(const Fe& key, const Fe& value) {  // (:745-800)
        auto p = path(key);
        auto pres = along(p);
        bool existed = !pres.back()[p.back()].is_zero();
        Fe cur = value;
        for (int lvl = height_ - 1; lvl >= 0; lvl--) {
            std::array<Fe, 8> pre = pres[lvl];
            pre[p[lvl]] = cur;
            cur = reg(pre);
        }
        root_ = cur;
        return existed;
    }

  private:
    Fe reg(const std::array<Fe, 8>& pre) {
        Fe h = cache_.hash8(pre);
        children_[h] = pre;
        return h;
    }
    std::vector<std::array<Fe, 8>> along(const std::vector<int>& p) const {
        std::vector<std::array<Fe, 8>> out;
        Fe node = root_;
        for (int lvl = 0; lvl < height_; lvl++) {
            auto it = children_.find(node);
            std::array<Fe, 8> pre;
            if (it != children_.end()) pre = it->second;
            else pre.fill(empty_root_for_height(height_ - lvl - 1));
            out.push_back(pre);
            node = pre[p[lvl]];
        }
        return out;
    }
    int field_, height_;
    PoseidonCache& cache_;
    std::map<Fe, std::array<Fe, 8>> children_;
    std::vector<Fe> empty_roots_;
    Fe root_;
};

struct Affine { Fe x, y; };       // Montgomery limbs, identity = (0,0)
struct Jacobian { Fe x, y, z; };  // Montgomery limbs, identity: z = 0

// Resident commitment key; commit(v) = sum v_i * ck_i over ck[..v.len()].
class CommitmentKey {
  public:
    CommitmentKey(int curve, const std::vector<Affine>& ck, bool precompute) : curve_(curve) {
        check(lurk_hip_msm_ctx_create(&ctx_, curve, ck.data(), ck.size(), precompute ? LURK_MSM_FLAG_PRECOMPUTE : 0));
    }
    ~CommitmentKey() { lurk_hip_msm_ctx_destroy(ctx_); }
    CommitmentKey(const CommitmentKey&) = delete;
    Jacobian commit(const std::vector<Fe>& scalars, bool is_mont) const {
        Jacobian out;
        check(lurk_hip_msm_ctx_run(ctx_, &out, scalars.data(), scalars.size(), is_mont ? 1 : 0));
        return out;
    }
    // canonical affine (x, y) of a commitment: what bit-exactness is defined on
    std::array<Fe, 2> to_affine(const Jacobian& p) const {
        std::array<Fe, 2> xy;
        check(lurk_hip_point_to_affine_canonical(curve_, xy.data(), &p));
        return xy;
    }

    // the key folded by the weights of k inner-product rounds at once: n / weights.size() affine points into device memory (window-table keys)
    void fold_key(size_t n, const std::vector<Fe>& weights_mont, void* d_out_affine64, void* stream = nullptr) const {
        check(lurk_hip_msm_ctx_fold_key_dev(ctx_, n, weights_mont.data(), weights_mont.size(), d_out_affine64, stream));
    }
    lurk_hip_msm_ctx* handle() const { return ctx_; }

  private:
    int curve_;
    lurk_hip_msm_ctx* ctx_ = nullptr;
};

// The same key cut across a list of devices and driven from this one process (arecibo's prover is a single process,
// /root/reference/src/proof/nova.rs:304-326): slice i of ck lives on devices[i], a commit runs the slices
// concurrently and sums the 96-byte partial commitments on the host.
class MultiCommitmentKey {
  public:
    MultiCommitmentKey(int curve, const std::vector<Affine>& ck, const std::vector<int>& devices, bool precompute) : curve_(curve) {
        check(lurk_hip_msm_multi_create(&ctx_, curve, ck.data(), ck.size(), devices.data(), (int)devices.size(),
                                        precompute ? LURK_MSM_FLAG_PRECOMPUTE : 0));
    }
    ~MultiCommitmentKey() { lurk_hip_msm_multi_destroy(ctx_); }
    MultiCommitmentKey(const MultiCommitmentKey&) = delete;
    Jacobian commit(const std::vector<Fe>& scalars, bool is_mont) const {
        Jacobian out;
        check(lurk_hip_msm_multi_commit(ctx_, &out, scalars.data(), scalars.size(), is_mont ? 1 : 0));
        return out;
    }
    int num_shards() const { return lurk_hip_msm_multi_num_shards(ctx_); }
    lurk_hip_msm_multi* handle() const { return ctx_; }
    // (device, first point, point count) of slice i
    std::array<size_t, 3> shard(int i) const {
        int dev = 0;
        size_t first = 0, count = 0;
        check(lurk_hip_msm_multi_shard(ctx_, i, &dev, &first, &count));
        return {(size_t)dev, first, count};
    }

  private:
    int curve_;
    lurk_hip_msm_multi* ctx_ = nullptr;
};

// R1CS shape resident on the GPU: the arithmetic of one folding step between its two commitments
// (arecibo R1CSShape::multiply_vec / commit_T's cross term / RelaxedR1CSWitness::fold, as reached from
// /root/reference/src/proof/nova.rs:291-293).  Matrices as arecibo's SparseMatrix {data, indices, indptr}.
struct SparseMatrix {
    std::vector<Fe> data;           // Montgomery values
    std::vector<uint64_t> indices;  // column into z = [W | u | X]
    std::vector<uint64_t> indptr;   // num_cons + 1
};

class R1CSShape {
  public:
    R1CSShape(int field, size_t num_cons, size_t num_vars, size_t num_io, const SparseMatrix& a, const SparseMatrix& b, const SparseMatrix& c)
        : field_(field), num_cons_(num_cons), num_vars_(num_vars), num_cols_(num_vars + 1 + num_io) {
        check(lurk_hip_r1cs_create(&h_, field, num_cons, num_vars, num_io, a.indptr.data(), a.indices.data(), a.data.data(), b.indptr.data(),
                                   b.indices.data(), b.data.data(), c.indptr.data(), c.indices.data(), c.data.data()));
    }
    ~R1CSShape() { lurk_hip_r1cs_destroy(h_); }
    R1CSShape(const R1CSShape&) = delete;
    // (A z, B z, C z)
    std::array<std::vector<Fe>, 3> multiply_vec(const std::vector<Fe>& z) const {
        if (z.size() != num_cols_) throw std::invalid_argument("z must have num_vars + 1 + num_io entries");
        std::array<std::vector<Fe>, 3> out{std::vector<Fe>(num_cons_), std::vector<Fe>(num_cons_), std::vector<Fe>(num_cons_)};
        check(lurk_hip_r1cs_multiply_vec(h_, z.data(), out[0].data(), out[1].data(), out[2].data()));
        return out;
    }
    // T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1
    std::vector<Fe> cross_term(const std::vector<Fe>& z1, const std::vector<Fe>& z2) const {
        if (z1.size() != num_cols_ || z2.size() != num_cols_) throw std::invalid_argument("z must have num_vars + 1 + num_io entries");
        std::vector<Fe> t(num_cons_);
        check(lurk_hip_r1cs_cross_term(h_, z1.data(), z2.data(), t.data()));
        return t;
    }
    // a + r b
    std::vector<Fe> fold(const std::vector<Fe>& a, const std::vector<Fe>& b, const Fe& r) const {
        if (a.size() != b.size()) throw std::invalid_argument("length mismatch");
        std::vector<Fe> out(a.size());
        check(lurk_hip_fold_vec(field_, a.data(), b.data(), &r, a.size(), out.data()));
        return out;
    }

    lurk_hip_r1cs* handle() const { return h_; }
    size_t num_cons() const { return num_cons_; }
    size_t num_cols() const { return num_cols_; }
    size_t num_vars() const { return num_vars_; }

  private:
    int field_;
    size_t num_cons_, num_vars_, num_cols_;
    lurk_hip_r1cs* h_ = nullptr;
};

// compute_witness_size (/root/reference/src/lem/multiframe.rs:503-516): elements a slot contributes to the witness vector
inline size_t slot_witness_size(int field_id, int slot_type) {
    size_t n = 0;
    check(lurk_hip_slot_witness_size(field_id, slot_type, &n));
    return n;
}
// allocate_slot's aux block for n slots of one type (circuit.rs:242-315), host buffers: n x size Montgomery elements
inline std::vector<Fe> slot_witness(int field_id, int slot_type, const std::vector<Fe>& preimages, bool preimages_mont) {
    const size_t per = slot_type == LURK_SLOT_BIT_DECOMP ? 1 : (size_t)slot_type, n = preimages.size() / per;
    std::vector<Fe> out(n * slot_witness_size(field_id, slot_type));
    check(lurk_hip_slot_witness(field_id, slot_type, preimages.data(), n, preimages_mont ? 1 : 0, out.data()));
    return out;
}
// StoreCore::hydrate_z_cache (/root/reference/src/lem/store_core.rs:256-269) over a topologically ordered node array
inline std::vector<Fe> store_hydrate(int field_id, const std::vector<lurk_hip_store_node>& nodes, const std::vector<Fe>& values, size_t* levels = nullptr) {
    std::vector<Fe> digests(nodes.size());
    check(lurk_hip_store_hydrate(field_id, nodes.data(), nodes.size(), values.data(), values.size(), digests.data(), levels));
    return digests;
}

// Store::to_scalar_vector (/root/reference/src/lem/store.rs:883-895): the public IO of a step, [tag, hash] per pointer
inline std::vector<Fe> to_scalar_vector(const std::vector<std::pair<Fe, Fe>>& z_ptrs) {
    std::vector<Fe> out;
    for (auto& zp : z_ptrs) { out.push_back(zp.first); out.push_back(zp.second); }
    return out;
}

// One curve's half of RecursiveSNARK::prove_step (/root/reference/src/proof/nova.rs:282-295): the running relaxed pair
// stays on the GPU; begin() returns the commitments the transcript absorbs, finish(r) folds.  The running instance's two
// commitments are folded here on the host as RelaxedR1CSInstance::fold does.
class FoldingContext {
  public:
    FoldingContext(int curve, R1CSShape& shape, CommitmentKey& key) : curve_(curve), shape_(shape) {
        check(lurk_hip_fold_ctx_create(&h_, curve, shape.handle(), key.handle()));
    }
    // the key cut across several devices: every commitment of the step runs its slices concurrently (no staging ahead in this form)
    FoldingContext(int curve, R1CSShape& shape, MultiCommitmentKey& key) : curve_(curve), shape_(shape) {
        check(lurk_hip_fold_ctx_create_multi(&h_, curve, shape.handle(), key.handle()));
    }
    ~FoldingContext() { lurk_hip_fold_ctx_destroy(h_); }
    FoldingContext(const FoldingContext&) = delete;
    std::array<Jacobian, 2> begin(const std::vector<Fe>& w2_mont, const std::vector<Fe>& x2_mont) {
        std::array<Jacobian, 2> c;
        check(lurk_hip_fold_step_begin(h_, w2_mont.data(), 0, nullptr, x2_mont.data(), &c[0], &c[1]));
        open_ = c;
        return c;
    }
    // called once per step from inside begin / begin_prefetched / step, after the step's device work has been enqueued and before the
    // call blocks on the commitments: where the next witness's slot traces are enqueued (lurk_hip_fold_ctx_set_submit_hook)
    void set_submit_hook(std::function<void()> fn) {
        hook_ = std::move(fn);
        check(lurk_hip_fold_ctx_set_submit_hook(h_, hook_ ? &FoldingContext::hook_trampoline : nullptr, this));
    }
    // staging ahead across devices: `helper` holds the same key on another device; staged commitments run on the helpers in turn
    void add_helper(CommitmentKey& helper) { check(lurk_hip_fold_ctx_add_helper(h_, helper.handle())); }
    // staging ahead: positions [offset, offset + range.size()) of the next fresh witness, its commitment starts now
    void prefetch(const std::vector<Fe>& w2_range_mont, size_t offset = 0) {
        check(lurk_hip_fold_step_prefetch(h_, w2_range_mont.data(), offset, w2_range_mont.size(), 0, nullptr));
    }
    // the step of the oldest staged instance; patches = (offset, values) ranges of W2 known only now
    std::array<Jacobian, 2> begin_prefetched(const std::vector<Fe>& x2_mont, const std::vector<std::pair<size_t, std::vector<Fe>>>& patches = {}) {
        std::vector<lurk_hip_w2_patch> ps;
        for (auto& p : patches) ps.push_back(lurk_hip_w2_patch{p.first, p.second.size(), p.second.data()});
        std::array<Jacobian, 2> c;
        check(lurk_hip_fold_step_begin_prefetched(h_, ps.data(), ps.size(), x2_mont.data(), &c[0], &c[1]));
        open_ = c;
        return c;
    }
    // folds the vectors on the device and the instance (comm_W1 + r comm_W2, comm_E1 + r comm_T, u1 + r, X1 + r X2) on the host,
    // both inside the library (RelaxedR1CSWitness::fold / RelaxedR1CSInstance::fold)
    void finish(const Fe& r_mont) {
        check(lurk_hip_fold_step_finish(h_, &r_mont));
        refresh();
    }
    // NIFS::prove whole: the challenge comes from the library's transcript (arecibo PoseidonRO over the other field of the cycle,
    // absorbing pp_digest, U1, U2, comm_T); returns {comm_W2, comm_T} and leaves r (Montgomery) in last_r
    std::array<Jacobian, 2> step(const std::vector<Fe>& w2_mont, const std::vector<Fe>& x2_mont, const Fe& pp_digest) {
        std::array<Jacobian, 2> c;
        check(lurk_hip_fold_step(h_, w2_mont.data(), 0, nullptr, x2_mont.data(), &pp_digest, &c[0], &c[1], &last_r));
        open_ = c;
        refresh();
        return c;
    }
    // the running instance's scalar part: u and X (Montgomery)
    std::pair<Fe, std::vector<Fe>> u_and_x() const {
        Fe u;
        std::vector<Fe> x(shape_.num_cols() - shape_.num_vars() - 1);
        check(lurk_hip_fold_ctx_instance(h_, nullptr, nullptr, &u, x.data()));
        return {u, x};
    }
    // host copies of z = [W | u | X] and E (Montgomery)
    std::pair<std::vector<Fe>, std::vector<Fe>> read() const {
        std::vector<Fe> z(shape_.num_cols()), e(shape_.num_cons());
        check(lurk_hip_fold_ctx_read(h_, z.data(), e.data()));
        return {z, e};
    }
    Jacobian comm_w{}, comm_e{};  // the running instance's commitments; identity (z = 0): RelaxedR1CSInstance::default
    Fe last_r{};                  // the challenge of the last step() (Montgomery)

  private:
    void refresh() { check(lurk_hip_fold_ctx_instance(h_, &comm_w, &comm_e, nullptr, nullptr)); }
    static int hook_trampoline(void* self) {
        try {
            static_cast<FoldingContext*>(self)->hook_();
            return 0;
        } catch (...) {  // (an exception must not unwind through the library's frames: the begin fails and is rolled back)
            return 1;
        }
    }
    std::function<void()> hook_;
    int curve_;
    R1CSShape& shape_;
    lurk_hip_fold_ctx* h_ = nullptr;
    std::array<Jacobian, 2> open_{};
};

// SuperNova's non-uniform IVC on one curve (/root/reference/src/proof/supernova.rs:226-244; the circuit of a step is chosen by
// its program counter, /root/reference/src/lem/multiframe.rs:271-356): one shape and one running pair per circuit index under ONE
// commitment key; a step folds into the running pair of its own circuit, the others are untouched.
class NivcFoldingContext {
  public:
    NivcFoldingContext(int curve, const std::vector<R1CSShape*>& shapes, CommitmentKey& key) {
        for (R1CSShape* sh : shapes) ctxs_.emplace_back(sh ? new FoldingContext(curve, *sh, key) : nullptr);
    }
    std::array<Jacobian, 2> step(size_t pc, const std::vector<Fe>& w2_mont, const std::vector<Fe>& x2_mont, const Fe& pp_digest) {
        if (pc >= ctxs_.size() || !ctxs_[pc]) throw std::invalid_argument("no circuit with this index");
        return ctxs_[pc]->step(w2_mont, x2_mont, pp_digest);
    }
    FoldingContext& circuit(size_t pc) { return *ctxs_.at(pc); }
    size_t num_circuits() const { return ctxs_.size(); }

  private:
    std::vector<std::unique_ptr<FoldingContext>> ctxs_;
};

// arecibo PoseidonRO (the transcript of NIFS::prove): absorb canonical elements of field_id, squeeze num_bits bits
inline Fe nova_ro_squeeze(int field_id, const std::vector<Fe>& elems, unsigned num_bits = 128) {
    Fe out;
    check(lurk_hip_nova_ro_squeeze(field_id, elems.data(), elems.size(), num_bits, &out));
    return out;
}

}  // namespace host
}  // namespace lurk

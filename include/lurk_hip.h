/*
 * lurk_hip.h - C ABI of liblurk_hip.so: the MI355X (gfx950) proving hot path for Lurk.
 *
 * This is the drop-in boundary.  lurk-beta has no backend plugin registry; the seam is one level
 * below its Rust generics, where arecibo calls the `pasta-msm` C symbols and where lurk-beta's
 * PoseidonCache calls neptune.  Every entry point below names the
 * reference interface it replaces.  Conventions kept from that seam:
 *   - plain pointers and sizes only; the caller owns every buffer, borrowed for the call;
 *   - field elements are 32 bytes, 4 x u64 little-endian limbs;
 *   - curve points use the pasta_curves `repr-c` layouts (/root/reference/Cargo.toml:42):
 *       affine  {x, y}      64 B, Montgomery form, identity = (0, 0)
 *       jacobian{x, y, z}   96 B, Montgomery form, identity has z = 0
 *   - scalars handed to the MSM are Montgomery form when is_mont != 0 (how pasta_curves stores
 *     them in memory), canonical integers otherwise;
 *   - Poseidon / NTT take and return canonical bytes (`PrimeField::to_repr()`,
 *     /root/reference/src/field.rs:72-75);
 *   - every function returns 0 on success, non-zero on failure; lurk_hip_last_error() then holds
 *     the message for the calling thread (pasta-msm returns {code, message}; the Rust side panics
 *     on non-zero).  There is NO CPU fallback: without a usable gfx950 device every compute entry
 *     point fails with LURK_HIP_ERR_NO_DEVICE.
 *   - entry points are thread-safe (arecibo commits from rayon worker threads).
 *   - *_dev variants take device pointers and a hipStream_t (as void*; NULL = default stream) and
 *     do not synchronise: inputs stay resident in HBM across calls.  Device buffers of field elements / points must be
 *     16-byte aligned (anything hipMalloc returns, at any multiple of 32 bytes); host buffers need no alignment.
 */
#ifndef LURK_HIP_H
#define LURK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* field ids (LurkField instances, /root/reference/src/field.rs:265-279) */
#define LURK_FIELD_PALLAS_FP 0 /* pasta_curves::Fp  = pallas::Base  = vesta::Scalar */
#define LURK_FIELD_PALLAS_FQ 1 /* pasta_curves::Fq  = pallas::Scalar = vesta::Base  */
#define LURK_FIELD_BN254_FR 2  /* halo2curves::bn256::Fr (the field of every KAT in the reference) */

/* curve ids (CurveCycleEquipped engines, /root/reference/src/proof/nova.rs:40-71) */
#define LURK_CURVE_PALLAS 0
#define LURK_CURVE_VESTA 1

#define LURK_HIP_OK 0
#define LURK_HIP_ERR_NO_DEVICE 1
#define LURK_HIP_ERR_INVALID_ARG 2
#define LURK_HIP_ERR_HIP 3
#define LURK_HIP_ERR_OOM 4

/* ---- runtime ---------------------------------------------------------------------------- */
int lurk_hip_device_count(void);
const char* lurk_hip_last_error(void);
const char* lurk_hip_version(void);
/* The ABI revision this header describes; lurk_hip_abi_version() is what the loaded library was built from (a binding compares the
 * two at start-up).  2 (round 6): lurk_hip_msm_ctx_info's *precomputed is a boolean (the key's form comes from lurk_hip_msm_ctx_form),
 * LURK_MSM_SLOTS is 6, LURK_MSM_SUBMIT_FOLLOW, the parameter blocks lurk_hip_ro_params / lurk_hip_ck_params, lurk_hip_scratch_trim.
 * 1: rounds 1-4 (*precomputed returned the form 0 / 1 / 2, four slots). */
#define LURK_HIP_ABI_VERSION 2
int lurk_hip_abi_version(void);
/* Prover scratch (the compressing SNARK's sum-check tables, eq tables, opening-argument halves) comes from one stack arena per
 * (device, stream) that grows by blocks of >= 256 MiB and is KEPT between proofs: about 2 GB per stream that has run a 2^20-row proof,
 * for the life of the process.  trim releases every arena no prover is using right now (nothing is live in it) back to the driver;
 * *released_bytes (may be NULL) = what went back.  The next proof on that stream re-allocates (a device-wide synchronisation the first
 * time a size is seen): call it when a process stops proving, not between proofs.
 * Threads: a prover call (lurk_hip_spartan_prove*_dev, lurk_hip_ipa_prove_dev, lurk_hip_sumcheck_prove*_dev) holds its stream's arena
 * for as long as the call runs - two calls on ONE stream from two threads run one after the other, calls on different streams run
 * side by side.  No entry point of the library takes the arenas of two streams at once, so the arenas cannot deadlock among
 * themselves; a caller that wraps prover calls in locks of its own must take them in one order. */
int lurk_hip_scratch_trim(size_t* released_bytes);
/* bind the calling thread to a device (one process per GPU; the multi-GPU layer calls this) */
int lurk_hip_set_device(int device);
/* per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) */
int lurk_hip_profile_enable(int on);
int lurk_hip_profile_reset(void);
/* total milliseconds and launch count recorded for kernels whose name starts with `prefix` */
int lurk_hip_profile_get(const char* prefix, double* total_ms, uint64_t* launches);

/* ---- Pedersen MSM ------------------------------------------------------------------------
 * Replaces pasta-msm's
 *   void mult_pippenger_pallas(pallas_point* out, const pallas_affine* points, size_t npoints,
 *                              const pallas_scalar* scalars, bool is_mont);        (and _vesta)
 * reached from arecibo CommitmentEngine::commit -> DlogGroup::vartime_multiscalar_mul, whose
 * lurk-beta callers are RecursiveSNARK::new / prove_step (/root/reference/src/proof/nova.rs:287-293,
 * /root/reference/src/proof/supernova.rs:231-244) and PublicParams::setup (nova.rs:205). */
int lurk_hip_msm_pallas(void* out_jacobian96, const void* bases_affine64, size_t npoints,
                        const void* scalars32, int is_mont);
int lurk_hip_msm_vesta(void* out_jacobian96, const void* bases_affine64, size_t npoints,
                       const void* scalars32, int is_mont);
/* The two symbols above under the names and the signature pasta-msm's src/lib.rs binds (pasta-msm 0.1.x, arecibo's
 * dependency; un-vendored: /root/reference/Cargo.toml:128 pulls it in through nova): a pasta-msm whose build script links
 * liblurk_hip.so instead of compiling its own C objects needs no source change.  They return nothing, as the originals; a
 * failure (no device, allocation) prints lurk_hip_last_error() and aborts the process. */
/* Opt-in key cache of the four one-shot symbols (default off; LURK_MSM_ONESHOT_KEY_CACHE=1 in the environment, read once when the library is loaded, turns it on for
 * callers that link mult_pippenger_* / cuda_pippenger_* without a source change and so cannot call this function):
 * arecibo commits under ONE immutable key at ONE address for a whole proof, yet the pasta-msm signature makes it hand the
 * bases over on every call (64 of the 96 bytes per point that cross PCIe).  With the cache on, the bases of the previous call
 * stay in HBM; a call with the same `points` pointer, npoints <= the cached length and bit-identical points at 4 096 sampled
 * positions reuses them (2^22: 14.0 ms -> 8.9 ms per call, of which 2.7 ms are the scalars' own upload).  A buffer rewritten in place at unsampled positions only would be missed:
 * do not enable it for callers that edit a key in place. */
int lurk_hip_msm_oneshot_key_cache(int enable);
#include <stdbool.h>
void mult_pippenger_pallas(void* out_jacobian96, const void* points_affine64, size_t npoints, const void* scalars32, bool is_mont);
void mult_pippenger_vesta(void* out_jacobian96, const void* points_affine64, size_t npoints, const void* scalars32, bool is_mont);
/* pasta-msm's GPU symbols (its `cuda` feature over sppark; what arecibo's GPU build binds instead of the two above, SURVEY.md
 * section 8b [MEM]): same arguments, sppark's RustError returned BY VALUE - code 0 and a NULL message on success; otherwise the
 * library's error code and a malloc'd message that the caller frees (sppark's Rust side does, in Drop).  No abort, no fallback. */
typedef struct lurk_hip_rust_error {
    int code;
    char* message;
} lurk_hip_rust_error;
lurk_hip_rust_error cuda_pippenger_pallas(void* out_jacobian96, const void* points_affine64, size_t npoints, const void* scalars32, bool is_mont);
lurk_hip_rust_error cuda_pippenger_vesta(void* out_jacobian96, const void* points_affine64, size_t npoints, const void* scalars32, bool is_mont);

/* Resident-bases context: the commitment key `ck` is constant for the whole proof
 * (/root/reference/src/proof/nova.rs:196-216), so it is uploaded once and kept in HBM.
 * Mirrors the msm-context API of the argumentcomputer pasta-msm fork (init(points) -> ctx,
 * with(ctx, scalars) -> point).  flags: bit0 = build the per-window precomputed table
 * (2^(c*k) * P_i for every window k; costs windows x 64 B x npoints of HBM; every window then shares
 * one bucket set, so the window grows to c = 20 bits from 2^19 points on: 13 n mixed additions instead of
 * 16 n); bits 8..15 = window-bit override for the table mode (16..20, 0 = automatic).
 * Footprint of bit0 for keys of <= 2^16 points with no window override: the SMALL form - every multiple m * 2^(c k) * P_i
 * resident, 256 KiB per point (2.6 GB at 10^4 points, 4.3 GB at 2^14, 5.6 GB at 2^16) plus <= 1 GiB of scratch while it is
 * built - so that a commitment is one launch (0.13 ms at 2^13 points instead of 0.35).  It is chosen only when it fits a quarter
 * of the device memory free at creation time; otherwise, and always with LURK_MSM_FLAG_WINDOW_BITS(16), the key takes the window
 * table (64 B x 16 per point).  lurk_hip_msm_ctx_form reports the form that was taken; a caller that needs the choice to be
 * the same on every run and device states it: LURK_MSM_FLAG_SMALL_FORM (the small form, or LURK_HIP_ERR_INVALID_ARG / _OOM) or
 * LURK_MSM_FLAG_NO_SMALL_FORM (the window table whatever is free). */
typedef struct lurk_hip_msm_ctx lurk_hip_msm_ctx;
#define LURK_MSM_FLAG_PRECOMPUTE 1
#define LURK_MSM_FLAG_WINDOW_BITS(c) (((c) & 0xff) << 8)
#define LURK_MSM_FLAG_SMALL_FORM (1 << 16)
#define LURK_MSM_FLAG_NO_SMALL_FORM (1 << 17)
/* lurk_hip_msm_multi_create only: cut the key over as many of the listed devices as leave every slice at least 2^20 points (the first k
 * of the list; LURK_MSM_MULTI_MIN_SLICE_LOG in the environment moves the threshold).  A slice pays a whole commitment's latency-bound
 * chain (~0.64 ms) around 0.83 ms of accumulation per 2^20 points: smaller slices cost more than they take off. */
#define LURK_MSM_FLAG_AUTO_SLICES (1 << 18)
int lurk_hip_msm_ctx_create(lurk_hip_msm_ctx** ctx, int curve, const void* bases_affine64,
                            size_t npoints, int flags);
/* same, bases already in device memory (borrowed for the lifetime of the ctx unless precomputed) */
int lurk_hip_msm_ctx_create_dev(lurk_hip_msm_ctx** ctx, int curve, const void* d_bases_affine64,
                                size_t npoints, int flags, void* stream);
/* commit to the first nscalars bases: out = sum_i scalars[i] * bases[i]  (nscalars <= npoints,
 * as CommitmentEngine::commit uses ck[..v.len()]) */
int lurk_hip_msm_ctx_run(lurk_hip_msm_ctx* ctx, void* out_jacobian96, const void* scalars32,
                         size_t nscalars, int is_mont);
/* device scalars in, 96-byte result written to *host* memory out_jacobian96 after the stream
 * has been synchronised by this call (the result is needed by the host-side transcript) */
int lurk_hip_msm_ctx_run_dev(lurk_hip_msm_ctx* ctx, void* out_jacobian96, const void* d_scalars32,
                             size_t nscalars, int is_mont, void* stream);
/* Asynchronous form: up to LURK_MSM_SLOTS commitments in flight per context, each with its own
 * workspace and stream - e.g. commit(W) and commit(T) of one folding step, or the commitments of
 * consecutive steps - so that the latency-bound tail of one overlaps the throughput-bound bucket
 * accumulation of the next.  submit enqueues (ordered after `stream`, the stream that produced the
 * scalars) and returns; wait blocks for that slot and writes the 96-byte result to host memory. */
#define LURK_MSM_SLOTS 6
int lurk_hip_msm_ctx_submit_dev(lurk_hip_msm_ctx* ctx, int slot, const void* d_scalars32,
                                size_t nscalars, int is_mont, void* stream);
/* The same with a scheduling class.  Commitments in flight share the integer VALU; a prover knows which one its next step
 * waits for.  FOREGROUND: the accumulation is the plain launch (three waves per SIMD, raised wave priority) - the commitment
 * the host is about to wait for (commit(T) of the open folding step).  BACKGROUND: the persistent one-wave-per-SIMD
 * accumulation at the lowest priority whatever the size - work staged ahead (commit(W2) of the next step), which fills the
 * issue slots the foreground commitment leaves during its sort and its bucket reduction.  DEFAULT = submit_dev (by size).
 * FOLLOW (round 6): work staged ahead that must only take what the open step's serial chain leaves - commit(W2 of the next
 * step).  Sort and plan run at once at the LOWEST wave priority; the accumulation (persistent, two waves per SIMD - LURK_MSM_FOLLOW_WGS -,
 * lowest priority) starts behind the ACCUMULATION of the latest FOREGROUND commitment in flight on this context and fills the window the
 * step leaves idle (commit(T)'s bucket reduction, the host's transcript, the folds, the next cross term); the tail (finalize, bucket
 * reduction) keeps the raised priority: the next step waits for this commitment too. */
#define LURK_MSM_SUBMIT_DEFAULT 0
#define LURK_MSM_SUBMIT_FOREGROUND 1
#define LURK_MSM_SUBMIT_BACKGROUND 2
#define LURK_MSM_SUBMIT_FOLLOW 3
int lurk_hip_msm_ctx_submit_dev_mode(lurk_hip_msm_ctx* ctx, int slot, const void* d_scalars32, size_t nscalars, int is_mont,
                                     void* stream, int mode);
int lurk_hip_msm_ctx_wait(lurk_hip_msm_ctx* ctx, int slot, void* out_jacobian96);
/* A PAIR of commitments with disjoint supports in one pass (the L and R of an inner-product-argument round under the resident key:
 * arecibo ipa_pc, /root/reference/src/proof/nova.rs:57-62): ONE scalar vector; the scalars whose index has bit `sel_bit` clear commit
 * to out_lo, the others to out_hi - two key spaces of one sort / accumulate / reduce instead of two commitments that each scan all
 * n scalars for half of them.  Window-table keys only (precompute flag, more than 2^16 points or a window-bit override). */
int lurk_hip_msm_ctx_submit_pair_dev(lurk_hip_msm_ctx* ctx, int slot, const void* d_scalars32, size_t nscalars, int is_mont, void* stream,
                                     int sel_bit);
int lurk_hip_msm_ctx_wait_pair(lurk_hip_msm_ctx* ctx, int slot, void* out_lo_jacobian96, void* out_hi_jacobian96);
int lurk_hip_msm_ctx_destroy(lurk_hip_msm_ctx* ctx);
/* Points the context at other device-resident bases (borrowed, plain key) and keeps its workspaces: a key that changes every
 * round (the folded key of the inner-product argument) without re-allocation. */
int lurk_hip_msm_ctx_rebind_dev(lurk_hip_msm_ctx* ctx, const void* d_bases_affine64, size_t npoints);
/* Workspaces (sort buffers, task partials, buckets: ~1 GiB per slot at 2^22 points) are allocated on a slot's first use; a
 * prover that wants no allocation inside its first steps reserves them up front for the largest commitment it will make. */
int lurk_hip_msm_ctx_reserve(lurk_hip_msm_ctx* ctx, size_t nscalars, int slots);
/* *precomputed is a boolean (1 = the key holds precomputed multiples, in either table form).  The FORM of the resident key comes
 * from lurk_hip_msm_ctx_form: 0 = plain (64 B/point), 1 = window table (the only form that commits a pair in one pass:
 * lurk_hip_msm_ctx_submit_pair_dev), 2 = the small-commitment multiples table (keys of <= 2^16 points). */
#define LURK_MSM_FORM_PLAIN 0
#define LURK_MSM_FORM_TABLE 1
#define LURK_MSM_FORM_SMALL 2
int lurk_hip_msm_ctx_info(const lurk_hip_msm_ctx* ctx, int* curve, size_t* npoints, int* window_bits, int* precomputed);
int lurk_hip_msm_ctx_form(const lurk_hip_msm_ctx* ctx, int* form);
int lurk_hip_msm_ctx_device(const lurk_hip_msm_ctx* ctx, int* device); /* the device the key is resident on */
/* Commitment-key generation: arecibo's CommitmentEngine::setup(label, n) = DlogGroup::from_label as
 * PublicParams::setup reaches it (/root/reference/src/proof/nova.rs:196-216): SHAKE256(label) squeezed 32 bytes per point (host:
 * the XOF is sequential), point i = pasta_curves hash_to_curve("from_uniform_bytes") of its bytes - BLAKE2b-512 expand_message_xmd,
 * simplified SWU on the 3-isogenous curve, the degree-3 isogeny - one point per lane on the device, affine Montgomery out.
 * from_label_dev writes npoints x 64 B to device memory (synchronises `stream`); ctx_from_label builds the resident context in
 * place, no host copy of the key ever exists; hash_to_curve_dev is the per-point map on caller-supplied 32-byte strings. */
int lurk_hip_shake256(const void* in, size_t in_len, void* out, size_t out_len); /* host-only helper (the XOF above) */
/* Run-time parameters of the restatement above (round 6): arecibo's from_label and pasta_curves' hash_to_curve are un-vendored
 * (/root/reference/Cargo.toml:128,131) and /root/reference holds no key bytes, so what was recalled from memory is a FIELD, not a
 * literal: the XOF, the bytes squeezed per point, and the three parts of the hash_to_curve domain-separation tag
 * DST = domain_prefix "-" curve_name suite.  Process-wide, set(NULL) restores the defaults (in brackets); from_label_dev /
 * ctx_from_label / from_label_host read them per call.  hash_to_curve_dev takes its own domain prefix and uses the rest.
 * from_label_host is the same map on the host (no device; <= 2^16 points, ~0.3 ms each): the first points of a key for a LURKDUMP
 * probe record (`python -m lurk_beta_amd.dump probe FILE`) or a CPU test. */
#define LURK_CK_XOF_SHAKE256 0
#define LURK_CK_XOF_SHAKE128 1
typedef struct lurk_hip_ck_params {
    uint32_t struct_size;       /* sizeof(lurk_hip_ck_params): get() fills it, set() checks it */
    uint32_t xof;               /* LURK_CK_XOF_* over the label [SHAKE256] */
    uint32_t bytes_per_point;   /* XOF bytes handed to hash_to_curve per point, 1..64 [32] */
    uint32_t reserved;          /* 0 */
    char domain_prefix[32];     /* hash_to_curve's domain prefix, NUL-terminated ["from_uniform_bytes"] */
    char curve_name_pallas[16]; /* CURVE_ID of Pallas in the tag ["pallas"] */
    char curve_name_vesta[16];  /* ["vesta"] */
    char suite[32];             /* the tag's tail ["_XMD:BLAKE2b_SSWU_RO_"] */
} lurk_hip_ck_params;
int lurk_hip_ck_params_get(lurk_hip_ck_params* out);
int lurk_hip_ck_params_set(const lurk_hip_ck_params* params);
int lurk_hip_ck_from_label_host(int curve, const void* label, size_t label_len, size_t npoints, void* out_affine64);
int lurk_hip_ck_hash_to_curve_dev(int curve, const char* domain_prefix, const void* d_uniform32, size_t n, void* d_out_affine64,
                                  void* stream);
int lurk_hip_ck_from_label_dev(int curve, const void* label, size_t label_len, size_t npoints, void* d_out_affine64,
                               void* stream);
int lurk_hip_msm_ctx_from_label(lurk_hip_msm_ctx** ctx, int curve, const void* label, size_t label_len, size_t npoints,
                                int flags);
/* Key files.  The reference caches its public parameters - mostly the commitment key - on disk and
 * maps them back (/root/reference/src/public_parameters/mod.rs:33-56, disk_cache.rs:69-77).  save writes the resident key as it
 * sits in HBM (64-byte header + 64-byte affine Montgomery records; with_table != 0 also the per-window multiples of a
 * precomputed context); load maps the file and copies it straight into device memory.  load's flags: LURK_MSM_FLAG_PRECOMPUTE
 * takes the file's table when it has one and rebuilds it on the device otherwise (one inversion per point: faster than
 * reading 13 x the bytes from disk unless the file is hot in the page cache). */
int lurk_hip_msm_ctx_save(const lurk_hip_msm_ctx* ctx, const char* path, int with_table);
int lurk_hip_msm_ctx_load(lurk_hip_msm_ctx** ctx, const char* path, int flags);

/* One process, several GPUs.  arecibo's prover is a single process (/root/reference/src/proof/nova.rs:304-326: one
 * witness-producer thread, rayon inside), so the multi-GPU form of the commitment is a context that owns a list
 * of devices: the key is cut into len(devices) contiguous slices (the first npoints % n_dev slices hold one more
 * point), every slice stays resident on its device, and a commit runs the slices concurrently (one host thread
 * per device inside the library), brings the 96-byte partial commitments back and sums them on the host.  A
 * device id may appear more than once (two slices on one GPU).  flags as for lurk_hip_msm_ctx_create.
 * commit takes the whole scalar vector in host memory; commit_dev takes one device pointer per slice (slice i's
 * scalars resident on slice i's device; entries of slices beyond nscalars are ignored). */
typedef struct lurk_hip_msm_multi lurk_hip_msm_multi;
int lurk_hip_msm_multi_create(lurk_hip_msm_multi** ctx, int curve, const void* bases_affine64, size_t npoints,
                              const int* devices, int n_dev, int flags);
int lurk_hip_msm_multi_num_shards(const lurk_hip_msm_multi* ctx);
int lurk_hip_msm_multi_shard(const lurk_hip_msm_multi* ctx, int index, int* device, size_t* first, size_t* count);
int lurk_hip_msm_multi_commit(lurk_hip_msm_multi* ctx, void* out_jacobian96, const void* scalars32, size_t nscalars,
                              int is_mont);
int lurk_hip_msm_multi_commit_dev(lurk_hip_msm_multi* ctx, void* out_jacobian96, const void* const* d_scalars32,
                                  size_t n_slices, size_t nscalars, int is_mont); /* n_slices must equal num_shards */
/* Asynchronous form (as lurk_hip_msm_ctx_submit_dev_mode / _wait, LURK_MSM_SLOTS slots): the slices of a commitment are submitted
 * from the calling thread and run concurrently on their devices; after_streams (or NULL) holds, per slice, the stream ON THAT
 * SLICE'S DEVICE that produced its scalars (e.g. the stream a peer copy into the device was enqueued on).  wait sums the partial
 * commitments.  Two slots give commit(W) and commit(T) of a folding step in flight on every device at once. */
int lurk_hip_msm_multi_submit_dev(lurk_hip_msm_multi* ctx, int slot, const void* const* d_scalars32, void* const* after_streams,
                                  size_t n_slices, size_t nscalars, int is_mont, int mode);
int lurk_hip_msm_multi_wait(lurk_hip_msm_multi* ctx, int slot, void* out_jacobian96);
int lurk_hip_msm_multi_destroy(lurk_hip_msm_multi* ctx);

/* Group helpers used by the multi-GPU gather (sum of per-rank partial commitments) and by tests:
 * out = sum of `count` Jacobian points (host memory, 96 B each). */
int lurk_hip_point_sum(int curve, void* out_jacobian96, const void* points_jacobian96, size_t count);
/* One process per GPU (a Rust host under MPI / its own launcher instead of one process with a device list): every rank commits its
 * slice of the vector under its slice of the key (lurk_hip_msm_ctx_*), the ranks all-gather the 96-byte partial commitments with
 * whatever transport they have (RCCL ncclAllGather on bytes, MPI_Allgather, a socket) into `gathered` (world x 96 B, rank order), and
 * THIS is the only library call the exchange needs: out = the commitment of the whole vector.  Host memory, no device touched; the
 * identity partial of a rank with an empty slice is handled.  (lurk_beta_amd/distributed.py drives exactly this from Python for the
 * tests and bench.py --gpus N: test plumbing, not a second ABI.) */
int lurk_hip_point_sum_gathered(int curve, void* out_jacobian96, const void* gathered_jacobian96, size_t world);
/* out = [scalar] point on the host (the transcript-side folding of commitments: comm_W1 + r comm_W2, comm_E1 + r comm_T) */
int lurk_hip_point_mul(int curve, void* out_jacobian96, const void* point_jacobian96, const void* scalar32, int is_mont);
/* Jacobian (Montgomery) -> canonical affine bytes (x, y), 64 B; identity -> all zero */
int lurk_hip_point_to_affine_canonical(int curve, void* out_xy64, const void* point_jacobian96);

/* ---- Poseidon ----------------------------------------------------------------------------
 * Replaces neptune's Poseidon::new_with_preimage(preimage, constants).hash() as called by
 * PoseidonCache::hash3/hash4/hash6/hash8 (/root/reference/src/hash.rs:180-204); constants are
 * PoseidonConstants::new() (standard strength, Merkle-tree domain tag; hash.rs:59-84).
 * preimages: n x arity x 32 B canonical; digests: n x 32 B canonical.  arity in {3,4,6,8}
 * (anything else fails, as HashArity::from panics, hash.rs:19-29). */
int lurk_hip_poseidon_batch(int field_id, int arity, const void* preimages, size_t n, void* digests);
int lurk_hip_poseidon_batch_dev(int field_id, int arity, const void* d_preimages, size_t n,
                                void* d_digests, void* stream);
/* Dense arity-8 tree (the dense analogue of coprocessor::trie::Trie,
 * /root/reference/src/coprocessor/trie/mod.rs:434-481): node = hash8(children in index order).
 * n_leaves must be a power of 8, >= 8.  levels_or_null (if given) receives every internal level,
 * level 1 first, root last: (n_leaves/8 + n_leaves/64 + ... + 1) x 32 B. */
int lurk_hip_poseidon_tree8(int field_id, const void* leaves, size_t n_leaves, void* root32,
                            void* levels_or_null);
/* device form: d_levels must hold (n_leaves-1)/7 elements (all internal levels, root last) */
int lurk_hip_poseidon_tree8_dev(int field_id, const void* d_leaves, size_t n_leaves, void* d_levels,
                                void* stream);
/* the constants the library generated (canonical): rc has (rf+rp)*(arity+1) elements, mds
 * (arity+1)^2; pass NULL to query sizes only */
int lurk_hip_poseidon_constants(int field_id, int arity, int* rf, int* rp, void* rc, void* mds);
/* n hashes on the HOST (no device needed): the same digests as lurk_hip_poseidon_batch, for callers that hash one preimage at a
 * time, as PoseidonCache::hash{3,4,6,8} does on a cache miss (/root/reference/src/hash.rs:180-204) - a single hash costs a host
 * core ~20 us and a kernel launch + copies several times that; batches belong on the device.  The store hydration below uses it
 * for DAG levels too narrow to be worth a kernel's dependency chain. */
int lurk_hip_poseidon_hash_host(int field_id, int arity, const void* preimages, size_t n, void* digests);

/* ---- store hydration -----------------------------------------------------------------------
 * Replaces the recursive, node-by-node hashing of StoreCore::hydrate_z_cache / hash_ptr
 * (/root/reference/src/lem/store_core.rs:256-269) with the StoreHasher preimage layouts
 * (/root/reference/src/lem/store.rs:29-78).  nodes: the DAG in topological order (children before parents); the digest of
 * every node is written to digests32 (n x 32 B canonical; an atom's digest is its value).  Every level is hashed on the
 * device - one preimage gather + one Poseidon batch per (level, arity) - and nothing crosses PCIe in between.
 * *levels (optional) receives the depth of the DAG.  Tags are the u16 tag values of /root/reference/src/tag.rs. */
#define LURK_NODE_ATOM 0    /* digest = values32[value]                                   (store_core.rs:204) */
#define LURK_NODE_TUPLE2 2  /* hash4(tag_a, h_a, tag_b, h_b)                              (store.rs:32-36) */
#define LURK_NODE_TUPLE3 3  /* hash6(...)                                                 (store.rs:37-49) */
#define LURK_NODE_TUPLE4 4  /* hash8(...)                                                 (store.rs:50-65) */
#define LURK_NODE_COMPACT 5 /* hash4(h_a, tag_b, h_b, h_c)                                (store.rs:75-77) */
#define LURK_NODE_COMM 6    /* hash3(values32[value] = secret, tag_a, h_a)                (store.rs:70-73) */
typedef struct {
    uint32_t kind;     /* LURK_NODE_* */
    uint32_t tag;      /* the tag a parent's preimage records for this node */
    uint32_t child[4]; /* indices of earlier nodes (tuple2: 2, tuple3: 3, tuple4: 4, compact: 3, comm: 1) */
    uint32_t value;    /* atom: index of its value; comm: index of the secret */
    uint32_t reserved;
} lurk_hip_store_node;
int lurk_hip_store_hydrate(int field_id, const lurk_hip_store_node* nodes, size_t n, const void* values32, size_t n_values,
                           void* digests32, size_t* levels);

/* ---- slot witnesses: the Poseidon / bit-decomposition part of the witness vector, produced on the device ------------
 * Replaces generate_slots_witnesses (/root/reference/src/lem/multiframe.rs:520-592), which runs allocate_slot
 * (/root/reference/src/lem/circuit.rs:242-315) on a WitnessCS per slot: neptune's circuit2::poseidon_hash_allocated for
 * the Hash4 / Hash6 / Hash8 / Commitment slots (circuit.rs:212-235) and bellpepper's to_bits_le_strict for the BitDecomp
 * slots (circuit.rs:236-238).  A slot's block in W is, in the circuit's allocation order,
 *     hash slots:  [preimage (arity) | l^2, l^4, l^5 + key for every S-box | digest]
 *     bit decomp:  [value | bits of the value from the top, with the AND chain closing every run of 1-bits of p - 1]
 * as 32-byte Montgomery values: about 85 % of the step circuit's W (7 808 of 9 119 aux per frame on BN254,
 * /root/reference/src/lem/eval.rs:1960-1966).  With these W is assembled in HBM - [globals | frame 0 | frame 1 | ...],
 * each frame = its slots' blocks back to back, then the rest of the frame's aux (multiframe.rs:699-702,
 * circuit.rs:1429-1433) - and handed to lurk_hip_msm_ctx_submit_dev / lurk_hip_r1cs_cross_term_dev without crossing PCIe.
 * slot_type: the arity for hash slots (LURK_SLOT_COMMITMENT = 3, HASH4 = 4, HASH6 = 6, HASH8 = 8) or LURK_SLOT_BIT_DECOMP. */
#define LURK_SLOT_BIT_DECOMP 1
#define LURK_SLOT_COMMITMENT 3
#define LURK_SLOT_HASH4 4
#define LURK_SLOT_HASH6 6
#define LURK_SLOT_HASH8 8
/* elements per slot block = compute_witness_size (multiframe.rs:503-516); host-only, needs no device */
int lurk_hip_slot_witness_size(int field_id, int slot_type, size_t* size);
/* n slots of one type: preimages n x (arity | 1) x 32 B (Montgomery if preimages_mont, else canonical); slot i's block is
 * written at element d_w[d_offsets[i]] (device array) or, with d_offsets == NULL, at d_w[first + i * stride] */
int lurk_hip_slot_witness_dev(int field_id, int slot_type, const void* d_preimages, size_t n, int preimages_mont, void* d_w,
                              const uint64_t* d_offsets, size_t first, size_t stride, void* stream);
/* host buffers in and out, blocks back to back (n x size x 32 B): small inputs and tests */
int lurk_hip_slot_witness(int field_id, int slot_type, const void* preimages, size_t n, int preimages_mont, void* w_out);
/* every slot block of a MultiFrame in one call: counts5 / d_preimages5 are indexed (hash4, hash6, hash8, commitment, bit_decomp)
 * - the order generate_slots_witnesses walks a frame's hints in (multiframe.rs:527-536) - with counts per frame and, per type,
 * num_frames * count preimages in frame-major order.  Frame f's slot blocks are written back to back from
 * d_w[first + f * frame_len].  The per-type launches run concurrently inside; ordered after and before `stream`. */
int lurk_hip_frames_witness_dev(int field_id, size_t num_frames, const size_t* counts5, const void* const* d_preimages5,
                                int preimages_mont, void* d_w, size_t first, size_t frame_len, void* stream);
/* places n_blocks packed blocks of block_len elements (host or device memory) at d_w[first + b * stride]: the globals and
 * the non-slot remainder of every frame, which the CPU synthesis still produces */
int lurk_hip_witness_blocks_dev(void* d_w, size_t first, size_t stride, const void* src, int src_on_host, size_t n_blocks,
                                size_t block_len, void* stream);

/* ---- NTT ---------------------------------------------------------------------------------
 * No reference counterpart: radix-2 NTT over a Pasta field, natural order in
 * and out, omega = 5^((p-1)/2^32)^(2^(32-log_n)); inverse includes the 1/n scaling. */
int lurk_hip_ntt(int field_id, void* inout, unsigned log_n, int inverse);
int lurk_hip_ntt_dev(int field_id, void* d_inout, unsigned log_n, int inverse, void* stream);

/* ---- relaxed-R1CS folding -----------------------------------------------
 * The arithmetic arecibo's NIFS::prove runs on the CPU between the two commitments of a folding step
 * (caller: RecursiveSNARK::prove_step, /root/reference/src/proof/nova.rs:291-293; arecibo is the
 * un-vendored `nova` dependency, /root/reference/Cargo.toml:128): with these the witness vectors
 * stay in HBM from commit(W) through commit(T) to the next step.
 * Shape = the three CSR matrices as arecibo's SparseMatrix {data, indices, indptr} holds them:
 * indptr (num_cons + 1) x usize, indices nnz x usize (column into z = [W | u | X], i.e.
 * num_vars + 1 + num_io columns), data nnz x 32 B Montgomery.  Copied to the device at creation
 * (coefficients de-duplicated into a dictionary); the host arrays are only borrowed for the call. */
typedef struct lurk_hip_r1cs lurk_hip_r1cs;
int lurk_hip_r1cs_create(lurk_hip_r1cs** shape, int field_id, size_t num_cons, size_t num_vars,
                         size_t num_io, const uint64_t* a_indptr, const uint64_t* a_indices,
                         const void* a_data, const uint64_t* b_indptr, const uint64_t* b_indices,
                         const void* b_data, const uint64_t* c_indptr, const uint64_t* c_indices,
                         const void* c_data);
int lurk_hip_r1cs_destroy(lurk_hip_r1cs* shape);
int lurk_hip_r1cs_dims(const lurk_hip_r1cs* shape, int* field_id, size_t* num_cons, size_t* num_vars, size_t* num_io);
int lurk_hip_r1cs_info(const lurk_hip_r1cs* shape, size_t* nnz_a, size_t* nnz_b, size_t* nnz_c,
                       size_t* distinct_coefficients);
int lurk_hip_r1cs_device(const lurk_hip_r1cs* shape, int* device); /* the device the shape is resident on */
/* R1CSShape::multiply_vec: (A z, B z, C z); d_z has num_vars + 1 + num_io elements, outputs num_cons */
int lurk_hip_r1cs_multiply_vec_dev(const lurk_hip_r1cs* shape, const void* d_z, void* d_az, void* d_bz,
                                   void* d_cz, void* stream);
/* R1CSShape::commit_T's vector: T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1 (u_i = z_i[num_vars]) */
int lurk_hip_r1cs_cross_term_dev(lurk_hip_r1cs* shape, const void* d_z1, const void* d_z2, void* d_t,
                                 void* stream);
/* The same vector with the running instance's products CACHED (round 6): A z1, B z1, C z1 fold linearly with the running pair
 * (A (z1 + r z2) = A z1 + r A z2), so a prover that keeps them resident gathers from z2 alone - half the gather chains of the call
 * above.  u1_32_mont: the running u (32 bytes, HOST memory: the caller folds scalars itself, u <- u + r u2).  Outputs T and (A z2,
 * B z2, C z2).  The fold of the cache rides in the call: given the PREVIOUS step's products d_*z2_prev and its challenge r_prev32_mont
 * (host; all four NULL = nothing to fold in), every cached row becomes row + r_prev * previous row - stored back in place - before it
 * is used, so that nothing stands between a step's transcript and the next cross term (two buffers: the previous products are read
 * while the new ones are written).  A folding context does all of this itself (lurk_hip_fold_step_*; LURK_FOLD_CACHED_PRODUCTS=0
 * keeps it on the call above). */
int lurk_hip_r1cs_cross_term_cached_dev(lurk_hip_r1cs* shape, const void* d_z2, void* d_az1, void* d_bz1, void* d_cz1, const void* u1_32_mont,
                                        const void* d_az2_prev, const void* d_bz2_prev, const void* d_cz2_prev, const void* r_prev32_mont, void* d_t,
                                        void* d_az2, void* d_bz2, void* d_cz2, void* stream);
/* RelaxedR1CSWitness::fold: out = a + r b over n elements (W1 + r W2, E1 + r T); r: 32 B Montgomery, host */
int lurk_hip_fold_vec_dev(int field_id, const void* d_a, const void* d_b, const void* r32_mont, size_t n,
                          void* d_out, void* stream);
/* count (<= 8) such folds under one r in ONE launch: d_a / d_b / d_out / n are host arrays of count device pointers / lengths
 * (out_k may be a_k: the fold is elementwise) - the step's finish(r) folds [W | u | X], E and the three cached products this way. */
int lurk_hip_fold_vecs_dev(int field_id, int count, const void* const* d_a, const void* const* d_b, const size_t* n, void* const* d_out,
                           const void* r32_mont, void* stream);
/* host-pointer forms of the three calls above (copy in, run, copy out): small inputs and tests */
int lurk_hip_r1cs_multiply_vec(const lurk_hip_r1cs* shape, const void* z, void* az, void* bz, void* cz);
int lurk_hip_r1cs_cross_term(lurk_hip_r1cs* shape, const void* z1, const void* z2, void* t);
int lurk_hip_fold_vec(int field_id, const void* a, const void* b, const void* r32_mont, size_t n, void* out);

/* ---- one folding step on one curve of the cycle ----------------------------------------------
 * What RecursiveSNARK::prove_step (/root/reference/src/proof/nova.rs:282-295, SuperNova /root/reference/src/proof/
 * supernova.rs:231-244) runs per curve through arecibo's NIFS::prove, with the running pair (z1 = [W1 | u1 | X1], E1)
 * resident in HBM from step to step.  A prover holds one context per curve: Pallas (primary, the Lurk step circuit) and
 * Vesta (secondary).  shape and key are borrowed (same curve / scalar field, created on the same device) and must outlive
 * the context; the context uses the key's async slots (W2: 0 and 2 alternately, T: 1, late ranges: 3).  The challenge r is a
 * Poseidon sponge over the OTHER field of the cycle absorbing the two commitments `begin` returns, hence two halves (a caller
 * with its own transcript uses them; lurk_hip_fold_step below runs begin, the library's own transcript and finish in one call):
 *   begin:  comm_W2 = commit(W2); T = cross term of (z1, [W2 | 1 | X2]); comm_T = commit(T)   - commitments in flight
 *   finish: W <- W1 + r W2, u <- u1 + r, X <- X1 + r X2, E <- E1 + r T                         - stream-ordered, no sync
 * The public IO X2 of a Lurk step is Store::to_scalar_vector's [tag, hash] x 3 (/root/reference/src/lem/store.rs:883-895,
 * Z1) in Montgomery form.  A new context holds the default (all-zero) relaxed pair, as RecursiveSNARK::new starts from. */
typedef struct lurk_hip_fold_ctx lurk_hip_fold_ctx;
int lurk_hip_fold_ctx_create(lurk_hip_fold_ctx** ctx, int curve, lurk_hip_r1cs* shape, lurk_hip_msm_ctx* key);
/* The same context over a key cut across several devices (lurk_hip_msm_multi_create; the north star's "witness-commitment batches
 * shard across the GPUs of one node"): the cross term and the folds run on the shape's device, each commitment pushes slice i of
 * its vector peer-to-peer into device i, the slices commit concurrently and the 96-byte partials are summed on the host.  begin /
 * finish / lurk_hip_fold_step as above; staging ahead (prefetch) is not offered with a multi-device key. */
int lurk_hip_fold_ctx_create_multi(lurk_hip_fold_ctx** ctx, int curve, lurk_hip_r1cs* shape, lurk_hip_msm_multi* key);
int lurk_hip_fold_ctx_destroy(lurk_hip_fold_ctx* ctx);
/* running pair <- host values (Montgomery): z1 (num_vars + 1 + num_io elements), E1 (num_cons elements) */
int lurk_hip_fold_ctx_set_running(lurk_hip_fold_ctx* ctx, const void* z1, const void* e1);
/* w2: num_vars x 32 B Montgomery, host memory or (w2_on_device) device memory produced on w2_stream (NULL = default
 * stream); x2_mont: num_io x 32 B, host */
int lurk_hip_fold_step_begin(lurk_hip_fold_ctx* ctx, const void* w2, int w2_on_device, void* w2_stream, const void* x2_mont,
                             void* comm_w2_jacobian96, void* comm_t_jacobian96);
/* Staging ahead (MI355X-side pipelining of nova.rs:282-326, where the witness of step i+1 is synthesized by a producer thread
 * while step i folds): the commitment to W2 needs no running instance, so it can run under the previous step's commit(T).
 *   prefetch(range)  stages positions [offset, offset + count) of the NEXT fresh witness (the rest zero for now) and starts
 *                    its commitment; at most two instances may be staged;
 *   begin_prefetched consumes the oldest staged instance: `patches` are the ranges known only now (the augmented circuit's
 *                    own variables around the step circuit's: they depend on the previous step's fold), host memory,
 *                    Montgomery; commit is linear, so comm_W2 = commit(staged ranges) + commit(late ranges).
 * The context uses the key's four async slots (lurk_hip_msm_ctx_reserve(key, n, 4)): W2 commitments alternate between slots 0
 * and 2, T uses slot 1, the late ranges slot 3 - or, up to 2^16 late positions, a small-commitment key of their own: the key's points
 * at those positions, built at the first begin that brings late ranges and kept while their layout stays the same (one launch per
 * step instead of a pass over a num_vars-long vector; LURK_FOLD_LATE_KEY=0 in the environment keeps slot 3).
 * lurk_hip_fold_step_begin is prefetch(whole W2) + begin_prefetched(no patches). */
typedef struct lurk_hip_w2_patch {
    size_t offset, count;   /* positions [offset, offset + count) of W2 */
    const void* values;     /* count x 32 bytes, host memory */
} lurk_hip_w2_patch;
int lurk_hip_fold_step_prefetch(lurk_hip_fold_ctx* ctx, const void* w2_range, size_t offset, size_t count, int on_device, void* stream);
/* Staging ahead ACROSS DEVICES: helper_key is the same commitment key resident on another device (an ordinary lurk_hip_msm_ctx over
 * the same bases, created under lurk_hip_set_device(other)).  Instances staged with lurk_hip_fold_step_prefetch then have their
 * commitment computed on the helpers in turn - the staged ranges are pushed peer-to-peer - while the context's own device folds the
 * open step; late ranges and T stay on the context's device.  Folding steps are sequential, so this (the reference's witness producer
 * thread, /root/reference/src/proof/nova.rs:306-317, spread over GPUs) is how IVC itself gains from several devices.  Add helpers
 * before the first step; they are borrowed for the context's lifetime and use their slots 0 and 2. */
int lurk_hip_fold_ctx_add_helper(lurk_hip_fold_ctx* ctx, lurk_hip_msm_ctx* helper_key);
int lurk_hip_fold_step_begin_prefetched(lurk_hip_fold_ctx* ctx, const lurk_hip_w2_patch* patches, size_t n_patches, const void* x2_mont,
                                        void* comm_w2_jac96, void* comm_t_jac96);
int lurk_hip_fold_step_finish(lurk_hip_fold_ctx* ctx, const void* r32_mont);
/* The challenge of the OPEN step from the library's transcript, for callers of the begin / finish halves who do not bring their own:
 * set the digest of the public parameters once; every begin then absorbs pp_digest and the running instance U1 - and runs the
 * permutation that completes - while the device is still working on the step, absorbs U2 when comm_W2 has arrived, and
 * lurk_hip_fold_step_challenge finishes behind comm_T with ONE permutation: r = RO(pp_digest, U1, U2, comm_T), the value
 * lurk_hip_nifs_challenge gives (and lurk_hip_fold_step uses the same staging).  r comes back in Montgomery form, ready for finish. */
int lurk_hip_fold_ctx_set_pp_digest(lurk_hip_fold_ctx* ctx, const void* pp_digest32);
int lurk_hip_fold_step_challenge(lurk_hip_fold_ctx* ctx, void* r32_mont);
/* A hook into the open step, for the caller's own device work that should run BESIDE the step - the slot traces of the next witness
 * (lurk_hip_slot_witness_dev), what the reference's witness-producer thread does while prove_step folds
 * (/root/reference/src/proof/nova.rs:304-326).  begin / begin_prefetched / lurk_hip_fold_step call hook(user) ONCE per step, on the
 * calling thread, after the step's device work (cross term, both commitments) has been enqueued and before they block on the
 * commitments.  Work enqueued from the hook queues behind the step's opening kernels and fills what the commitments leave (their
 * sorts and bucket reductions); enqueued before begin it would run first and hold the cross term back, enqueued after begin has
 * returned it runs alone: 3.50-3.59 ms per step at rc = 100 against 4.22 and 3.85 (profiles/r04_step_witness_placement.txt).
 * Of this context's own functions the hook may call lurk_hip_fold_step_prefetch only (the next instance is then staged AND its
 * commitment started in the background class, beside the open step's commit(T)); a non-zero return fails the begin (the step is rolled
 * back).  NULL removes it. */
typedef int (*lurk_hip_fold_submit_hook_fn)(void* user);
int lurk_hip_fold_ctx_set_submit_hook(lurk_hip_fold_ctx* ctx, lurk_hip_fold_submit_hook_fn hook, void* user);
/* the running pair where it lives (valid until the next finish) and the stream its updates are ordered on */
int lurk_hip_fold_ctx_running_dev(lurk_hip_fold_ctx* ctx, void** d_z, void** d_e, void** stream);
/* copies of the running pair for the host (either may be NULL); synchronises the context's stream */
int lurk_hip_fold_ctx_read(lurk_hip_fold_ctx* ctx, void* z_host, void* e_host);
/* The running INSTANCE U = (comm_W, comm_E, u, X) lives in the context beside the running witness: finish(r) folds it on the host
 * as RelaxedR1CSInstance::fold does (comm_W1 + r comm_W2, comm_E1 + r comm_T, u1 + r, X1 + r X2) while the device folds the
 * vectors.  set_instance installs the two commitments of a running pair given with set_running (u and X are taken from its z1);
 * instance reads the four parts back (96-byte Jacobians, Montgomery scalars; any pointer may be NULL). */
int lurk_hip_fold_ctx_set_instance(lurk_hip_fold_ctx* ctx, const void* comm_w_jacobian96, const void* comm_e_jacobian96);
int lurk_hip_fold_ctx_instance(lurk_hip_fold_ctx* ctx, void* comm_w_jacobian96, void* comm_e_jacobian96, void* u32_mont, void* x_mont);
/* NIFS::prove whole (/root/reference/src/proof/nova.rs:291-293 -> arecibo nifs.rs): begin, the challenge
 * r = RO(pp_digest, U1, U2, comm_T) derived in the library (lurk_hip_nifs_challenge below), finish(r).  pp_digest32: the
 * public parameters' digest, a canonical scalar (32 B).  Outputs (any may be NULL): the step's two commitments and r (Montgomery). */
int lurk_hip_fold_step(lurk_hip_fold_ctx* ctx, const void* w2, int w2_on_device, void* w2_stream, const void* x2_mont,
                       const void* pp_digest32, void* comm_w2_jacobian96, void* comm_t_jacobian96, void* r32_mont);

/* ---- the transcript: Nova's random oracle (host code, no device needed; restated [MEM], parity unpinned) ------------------
 * arecibo PoseidonRO over neptune's sponge API (simplex, arity 24, standard strength):
 *   nova_ro_squeeze     absorbs n canonical elements of `field_id` (32 B each), squeezes one element and returns the integer of
 *                       its low num_bits bits (32 B little-endian) - PoseidonRO::{absorb, squeeze};
 *   nova_ro_pattern_tag neptune IOPattern([Absorb(a), Squeeze(s)]).value(domain_separator): the sponge's capacity element (16 B);
 *   nifs_challenge      the absorb list of NIFS::prove on `curve`: pp_digest, U1 = (comm_W1, comm_E1, u1, X1 as 4 x 64-bit limbs
 *                       each), U2 = (comm_W2, X2), comm_T - commitments as (x, y, is_infinity), scalars through scalar_as_base -
 *                       then r = squeeze(NUM_CHALLENGE_BITS = 128) in Montgomery form of the curve's scalar field. */
/* Run-time parameters of the restatement above (round 6).  arecibo and neptune are un-vendored dependencies of the reference
 * (/root/reference/Cargo.toml:127-128) and /root/reference holds no transcript value, so every constant of the sponge and of the
 * absorb list that was recalled from memory is a FIELD here instead of a compiled-in literal: the first Rust-side run that gets
 * another r than arecibo's NIFS::prove (caller: /root/reference/src/proof/nova.rs:282-295) can move one field at a time - or hand a
 * LURKDUMP probe record (lurk_beta_amd/dump.py, rust/lurk-hip-sys/src/dump.rs) to `python -m lurk_beta_amd.dump probe FILE`, which
 * searches them - with no rebuild.  Process-wide; set(NULL) restores the defaults (the values of rounds 1-5, in brackets); a folding
 * context reads them when a step's transcript begins.  nova_ro_squeeze / nova_ro_pattern_tag use arity, domain_separator (squeeze
 * only), absorb_tag_bit, pattern_absorbs and squeeze_element; nifs_challenge and the folding contexts use all of them. */
#define LURK_RO_ITEM_PP_DIGEST 0
#define LURK_RO_ITEM_U1 1
#define LURK_RO_ITEM_U2 2
#define LURK_RO_ITEM_COMM_T 3
#define LURK_RO_PART_COMM_W 0
#define LURK_RO_PART_COMM_E 1
#define LURK_RO_PART_U 2
#define LURK_RO_PART_X 3
typedef struct lurk_hip_ro_params {
    uint32_t struct_size;        /* sizeof(lurk_hip_ro_params): get() fills it, set() checks it */
    uint32_t arity;              /* rate of the sponge = neptune arity, width arity + 1 [24] */
    uint32_t domain_separator;   /* IOPattern::value(domain_separator) [0] */
    uint32_t absorb_tag_bit;     /* SpongeOp::Absorb(n).value() = n + 2^absorb_tag_bit [31] */
    uint32_t num_challenge_bits; /* NUM_CHALLENGE_BITS: low bits of the squeezed element kept as r [128] */
    uint32_t item_order[4];      /* NIFS::prove absorbs LURK_RO_ITEM_* in this order [pp_digest, U1, U2, comm_T] */
    uint32_t relaxed_order[4];   /* RelaxedR1CSInstance::absorb_in_ro: LURK_RO_PART_* in this order [comm_W, comm_E, u, X] */
    uint32_t fresh_order[2];     /* R1CSInstance::absorb_in_ro: 0 = comm_W, 1 = X [comm_W, X] */
    uint32_t point_elements;     /* a commitment is (x, y, is_infinity) = 3 elements, or (x, y) = 2 [3] */
    uint32_t relaxed_x_limbs;    /* every X_i of a relaxed instance as this many limbs (BN_N_LIMBS); 0 = one element through scalar_as_base [4] */
    uint32_t fresh_x_limbs;      /* the same for a fresh instance [0] */
    uint32_t limb_bits;          /* BN_LIMB_WIDTH [64] */
    uint32_t pattern_absorbs;    /* 0 = the IO pattern declares the number of elements really absorbed; else this constant (NUM_FE_FOR_RO) [0] */
    uint32_t squeeze_element;    /* which rate element squeeze reads [0] */
} lurk_hip_ro_params;
int lurk_hip_ro_params_get(lurk_hip_ro_params* out);
int lurk_hip_ro_params_set(const lurk_hip_ro_params* params);
/* Diagnostic twin of nifs_challenge: the elements it absorbs, in order, canonical 32-byte integers of the RO's field (the other
 * field of the cycle) - what a probe record's `absorbed` list is compared with, element by element.  *count = the number of
 * elements; out_elems32 (cap elements) may be NULL to ask for the count only. */
int lurk_hip_nifs_absorb_list(int curve, const void* pp_digest32, const void* comm_w1_jacobian96, const void* comm_e1_jacobian96,
                              const void* u1_mont, const void* x1_mont, const void* comm_w2_jacobian96, const void* x2_mont, size_t num_io,
                              const void* comm_t_jacobian96, void* out_elems32, size_t cap, size_t* count);
int lurk_hip_nova_ro_squeeze(int field_id, const void* elems32, size_t n, unsigned num_bits, void* out32);
int lurk_hip_nova_ro_pattern_tag(uint32_t absorbs, uint32_t squeezes, uint32_t domain_separator, void* out16);
int lurk_hip_nifs_challenge(int curve, const void* pp_digest32, const void* comm_w1_jacobian96, const void* comm_e1_jacobian96,
                            const void* u1_mont, const void* x1_mont, const void* comm_w2_jacobian96, const void* x2_mont, size_t num_io,
                            const void* comm_t_jacobian96, void* r32_mont);

/* ---- arecibo's Keccak256Transcript (host code; no device needed) --------------------------------------------------------------
 * The transcript RelaxedR1CSSNARK / BatchedRelaxedR1CSSNARK run behind CompressedSNARK::prove (/root/reference/src/proof/nova.rs:92,
 * 341-356; supernova.rs:110, 293-302): persona tag "NoTR", domain-separator tag "NoDS", a 64-byte state, a u16 round counter and a
 * running Keccak-256 hasher; a challenge is Scalar::from_uniform of 64 squeezed bytes.  Restated from arecibo's published source
 * and UNPINNED (arecibo is an un-vendored dependency, /root/reference/Cargo.toml:128); Keccak-256 itself is pinned by known answers.
 * absorb_scalars takes canonical little-endian 32-byte elements and absorbs their to_transcript_bytes (the repr reversed);
 * absorb_point takes a 96-byte Jacobian and absorbs x || y || [finite] of its affine form. */
typedef struct lurk_hip_keccak_transcript lurk_hip_keccak_transcript;
int lurk_hip_keccak256(const void* in, size_t len, void* out32);
int lurk_hip_keccak_transcript_new(lurk_hip_keccak_transcript** t, const void* label, size_t label_len);
int lurk_hip_keccak_transcript_destroy(lurk_hip_keccak_transcript* t);
int lurk_hip_keccak_transcript_absorb(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, const void* bytes, size_t len);
int lurk_hip_keccak_transcript_absorb_scalars(lurk_hip_keccak_transcript* t, const void* label, size_t label_len,
                                              const void* scalars32_canonical, size_t n);
int lurk_hip_keccak_transcript_absorb_point(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, int curve,
                                            const void* point_jacobian96);
int lurk_hip_keccak_transcript_dom_sep(lurk_hip_keccak_transcript* t, const void* bytes, size_t len);
int lurk_hip_keccak_transcript_squeeze(lurk_hip_keccak_transcript* t, const void* label, size_t label_len, int field_id,
                                       void* out32_canonical);
/* Ready-made `challenge` arguments for the round loops below (lurk_hip_sumcheck_prove_dev / _prove_batch_dev, lurk_hip_ipa_prove_dev)
 * over a Keccak transcript, so that a proof's rounds run without leaving the library: pass the function as `challenge` and a
 * lurk_hip_keccak_round_binding as `user`.  Sum-check: the round polynomial's n_scalars coefficients are absorbed under absorb_label,
 * the challenge is squeezed under squeeze_label.  Inner-product argument: L under absorb_label, R under absorb_label2 (points of
 * `curve`), then the squeeze.  Every challenge is also appended to challenges_out (32 canonical bytes each; NULL = not kept), n_rounds
 * counts the calls. */
typedef struct lurk_hip_keccak_round_binding {
    lurk_hip_keccak_transcript* transcript;
    int field_id, n_scalars, curve;
    const void* absorb_label;
    size_t absorb_label_len;
    const void* absorb_label2;
    size_t absorb_label2_len;
    const void* squeeze_label;
    size_t squeeze_label_len;
    void* challenges_out;
    size_t challenges_cap, n_rounds; /* capacity of challenges_out in challenges; calls so far */
} lurk_hip_keccak_round_binding;
int lurk_hip_keccak_sumcheck_challenge(void* binding, int round, const void* coefficients32_canonical, void* out_r32_canonical);
int lurk_hip_keccak_ipa_challenge(void* binding, int round, const void* l_jacobian96, const void* r_jacobian96, void* out_r32_canonical);

/* ---- sum-check rounds (the data-parallel half of CompressedSNARK::prove) -----------------------
 * CompressedSNARK::prove (/root/reference/src/proof/nova.rs:341-356, supernova.rs:293-302) -> arecibo RelaxedR1CSSNARK::prove ->
 * SumcheckProof::prove_cubic_with_additive_term (outer: eq(tau) (Az Bz - (u Cz + E))) and prove_quad (inner: poly_ABC z).
 * The tables (Montgomery, len = 2^k elements each, device memory) stay in HBM, the transcript stays on the host.  One call is
 * one round: if bind_r32_mont != NULL every table's top variable is first bound to it in place (P[i] += r (P[len/2 + i] - P[i]),
 * the table is len / 2 long afterwards); then, if evals_out != NULL, the evaluations of the next round polynomial over the
 * current tables are summed and written to HOST memory (the call synchronises the stream for them):
 *   degree 3: d_polys = {A, B, C, D}, evals_out = [e(0), e(2), e(3)] of sum_i A (B C - D)     (3 x 32 B)
 *   degree 2: d_polys = {A, B},       evals_out = [e(0), e(2)]       of sum_i A B             (2 x 32 B)
 * A prover calls it with (NULL, evals) for the first round, (r_j, evals) in between and (r_last, NULL) at the end, after which
 * P[0] of every table is its evaluation at the challenge point.  (A whole sum-check is cheaper through lurk_hip_sumcheck_prove_dev /
 * _prove_batch_dev below: they keep the rounds' scratch - workgroup partial sums, ticket counter, the pinned words the totals land in -
 * for the whole proof, this entry point sets it up per call.) */
int lurk_hip_sumcheck_round_dev(int field_id, int degree, void* const* d_polys, size_t len, const void* bind_r32_mont,
                                void* evals_out, void* stream);
/* EqPolynomial::evals: d_out[b] = prod_j (b_j ? r_j : 1 - r_j), r_0 <-> the most significant bit of b; r: ell x 32 B Montgomery, host */
/* A whole sum-check (SumcheckProof::prove_quad / prove_cubic_with_additive_term of arecibo's Spartan, behind
 * /root/reference/src/proof/nova.rs:341-356) with the tables resident: log2(len) rounds of lurk_hip_sumcheck_round_dev, the round
 * polynomial interpolated on the host, the transcript behind a callback: `challenge` receives the round and the polynomial's
 * degree + 1 canonical coefficients (low to high) and writes r as a canonical 32-byte value (return 0; anything else aborts the call).
 * The tables are consumed: their contents are UNSPECIFIED after the call (the first rounds bind them in place; once they are down to
 * 2^LURK_SUMCHECK_HOST_TAIL_LOG elements the remaining rounds run on a host copy, so d_polys[k][0] is NOT the final evaluation) -
 * out_finals is the only place the final evaluations are returned.  Outputs, all canonical: the round polynomials (rounds x (degree + 1) x 32 bytes), the
 * tables' final evaluations (2 or 4 x 32 bytes) and the final claim. */
typedef int (*lurk_hip_sumcheck_challenge_fn)(void* user, int round, const void* coefficients32_canonical, void* out_r32_canonical);
int lurk_hip_sumcheck_prove_dev(int field_id, int degree, void* const* d_polys, size_t len, const void* claim32_canonical,
                                lurk_hip_sumcheck_challenge_fn challenge, void* user, void* out_polys, void* out_finals,
                                void* out_claim32, void* stream);
/* Batched form: sum_i coeff_i * sum_x comb(tables of instance i) with ONE challenge per round shared by all instances - the
 * evaluation-claim batching of arecibo's RelaxedR1CSSNARK and the outer / inner sum-checks of its BatchedRelaxedR1CSSNARK (SuperNova's
 * compressor, /root/reference/src/proof/supernova.rs:110, 293-302).  d_polys: n_instances x (4 or 2) device tables, instance-major, all
 * of length len (shorter instances zero-padded by the caller); coeffs32_canonical: n_instances batching coefficients; claim: the
 * COMBINED claim sum_i coeff_i claim_i.  out_finals: n_instances x (4 or 2) x 32 B, the tables' final evaluations, instance-major. */
int lurk_hip_sumcheck_prove_batch_dev(int field_id, int degree, size_t n_instances, void* const* d_polys, size_t len,
                                      const void* coeffs32_canonical, const void* claim32_canonical,
                                      lurk_hip_sumcheck_challenge_fn challenge, void* user, void* out_polys, void* out_finals,
                                      void* out_claim32, void* stream);
int lurk_hip_eq_evals_dev(int field_id, const void* r32_mont, int ell, void* d_out, void* stream);

/* ---- inner-product argument rounds (the opening half of CompressedSNARK::prove) ------------------
 * EE1 / EE2 = ipa_pc::EvaluationEngine on the Pasta cycle (/root/reference/src/proof/nova.rs:57-62).  Per round of the published
 * argument, with the vectors a, b and the (folded) key resident in HBM: the two cross inner products, the two commitments
 * L, R (lurk_hip_msm_ctx_create_dev / run_dev over the halves of the device key), then with the transcript's challenge r the
 * folds a' = r a_L + r^-1 a_R, b' = r^-1 b_L + r b_R (fold_halves, in place: the vector is len / 2 long afterwards) and
 * ck' = [r^-1] ck_L + [r] ck_R (points_fold_halves: len / 2 affine points out; may not alias the input). */
int lurk_hip_inner_product_dev(int field_id, const void* d_a, const void* d_b, size_t n, void* out32_mont, void* stream);
int lurk_hip_fold_halves_dev(int field_id, void* d_v, size_t len, const void* s_lo32_mont, const void* s_hi32_mont, void* stream);
int lurk_hip_points_fold_halves_dev(int curve, const void* d_points_affine64, size_t len, const void* s_lo32_mont,
                                    const void* s_hi32_mont, void* d_out_affine64, void* stream);

/* The same rounds WITHOUT folding the key (the form lurk_beta_amd/ipa.py uses when the prover's resident table key is at hand):
 * the folded key of round k is a fixed linear image of the original one, ck_k[p] = sum_{i = p mod m} coef_k[i] ck[i] (m = the
 * current length, coef_k[i] = the product of the fold weights position i met so far), so L and R are ordinary commitments under
 * the ORIGINAL key of the two length-n vectors round_scalars writes (n/2 non-zeros each):
 *   out_l[i] = a[(i mod m) - m/2] coef[i] if (i mod m) >= m/2 else 0;  out_r[i] = a[(i mod m) + m/2] coef[i] if (i mod m) < m/2 else 0
 * and after the challenge coef_fold multiplies coef[i] by s_lo / s_hi according to the half (i mod m) lies in (coef starts as
 * all ones; after the last round it is the verifier's s vector: the final key element is commit(ck, coef)).  Two table-mode
 * MSMs per round replace m/2 full-size double scalar multiplications whose 255-step ladder is latency-bound for every m. */
/* (d_out_r == NULL: ONE merged vector in d_out_l - L's and R's supports are disjoint - for lurk_hip_msm_ctx_submit_pair_dev with
 *  sel_bit = log2(m / 2): out_hi = L, out_lo = R) */
int lurk_hip_ipa_round_scalars_dev(int field_id, const void* d_a, size_t m, const void* d_coef, size_t n, void* d_out_l,
                                   void* d_out_r, void* stream);
int lurk_hip_ipa_coef_fold_dev(int field_id, void* d_coef, size_t n, size_t m, const void* s_lo32_mont, const void* s_hi32_mont,
                               void* stream);
/* The whole inner-product argument under a resident key (arecibo ipa_pc::InnerProductArgument::prove, the opening of CompressedSNARK on
 * the Pasta cycle: /root/reference/src/proof/nova.rs:57-62, 341-356): log2(n) rounds of { composed scalars, commitment of L and R under
 * the ORIGINAL key, cross inner products, challenge, folds } with the vectors resident; the transcript is the caller's: `challenge`
 * receives the round number and L, R (96-byte Jacobians) and writes r as a canonical 32-byte value below the group order (return 0;
 * anything else aborts the call).  d_a, d_b: n Montgomery scalars on the device, consumed (folded in place); ck_c: the extra base,
 * already scaled by the transcript's first challenge.  Outputs: L and R per round (log2(n) x 96 bytes each), a_hat canonical, and the
 * folded key element as an affine Montgomery point ((0, 0) = identity).  n: a power of two, at most the key's points. */
/* The key folded by the weights of k rounds at once: d_out[p] = sum_{b < n_weights} weights[b] * key[b m + p], m = n / n_weights affine
 * Montgomery points on the device ((0, 0) = identity) - what k rounds of ck' = [r^-1] ck_L + [r] ck_R leave, the weights being the
 * products of the rounds' fold weights by block (arecibo ipa_pc.rs; /root/reference/src/proof/nova.rs:57-62).  The key must be in the
 * window-table form; weights: host memory, Montgomery, n_weights a power of two <= 2^12; n a multiple of it, at most the key's points.
 * lurk_hip_ipa_prove_dev does this by itself after its fourth round under a key of >= 2^18 points and continues under the folded key
 * (LURK_IPA_FOLD_MIN_LOG in the environment moves the threshold, 0 = never): 2^20 elements in 21 ms instead of 39. */
int lurk_hip_msm_ctx_fold_key_dev(lurk_hip_msm_ctx* key, size_t n, const void* weights32_mont, size_t n_weights, void* d_out_affine64,
                                  void* stream);
typedef int (*lurk_hip_ipa_challenge_fn)(void* user, int round, const void* l_jacobian96, const void* r_jacobian96, void* out_r32_canonical);
int lurk_hip_ipa_prove_dev(lurk_hip_msm_ctx* key, void* d_a32, void* d_b32, size_t n, const void* ck_c_jacobian96,
                           lurk_hip_ipa_challenge_fn challenge, void* user, void* out_l_jacobian96, void* out_r_jacobian96,
                           void* out_a_hat32, void* out_ck_hat_affine64, void* stream);

/* The single-instance compressing prover as ONE call (what CompressedSNARK::prove runs per curve: /root/reference/src/proof/nova.rs:341-356
 * -> arecibo RelaxedR1CSSNARK::prove): outer cubic sum-check, inner quadratic sum-check over the transposed shape, the two evaluation claims
 * batched to one point, one inner-product argument under the resident key - a sequence of the entry points above with the vectors resident
 * and the Keccak transcript inside.  The protocol is this repository's own (oracle/spartan_ref.py, oracle/spartan_fast.py: the oracle's
 * prover gives the same proof element for element, its verifier accepts it), not arecibo's byte for byte.
 *   shape: num_cons x (num_vars + 1 + num_io) columns of z = [W | u | X]; shape_t: its transpose over 2 num_vars rows and num_cons columns
 *   (created with num_vars' = num_cons - 1, num_io' = 0); num_cons, num_vars: powers of two.  key: >= max(num_cons, num_vars) points, the
 *   key that committed W and E; ck_c: the inner-product base (a 96-byte Jacobian).  x, u: canonical; d_w (num_vars), d_e (num_cons):
 *   Montgomery, device memory, not modified.  label: the transcript's label.  Outputs (host memory, caller-sized, canonical):
 *   polys_outer log2(num_cons) x 4 x 32 B, claims_outer 3 x 32 (Az, Bz, Cz), eval_e 32, polys_inner (log2(num_vars) + 1) x 3 x 32,
 *   eval_w 32, polys_batch log2(N) x 3 x 32 (N = max(num_cons, num_vars)), evals_batch 2 x 32, ipa_l / ipa_r log2(N) x 96 (Jacobians),
 *   ipa_a 32. */
typedef struct lurk_hip_spartan_proof {
    void* polys_outer;
    void* claims_outer;
    void* eval_e;
    void* polys_inner;
    void* eval_w;
    void* polys_batch;
    void* evals_batch;
    void* ipa_l;
    void* ipa_r;
    void* ipa_a;
} lurk_hip_spartan_proof;
int lurk_hip_spartan_prove_dev(const lurk_hip_r1cs* shape, const lurk_hip_r1cs* shape_t, size_t num_cons, size_t num_vars, size_t num_io,
                               lurk_hip_msm_ctx* key, const void* ck_c_jacobian96, const void* x32_canonical, const void* u32_canonical,
                               const void* d_w32_mont, const void* d_e32_mont, const void* comm_w_jacobian96, const void* comm_e_jacobian96,
                               const void* label, size_t label_len, lurk_hip_spartan_proof* out, void* stream);

/* The BATCHED compressing prover as one call: n relaxed instances of different shapes and sizes under one key, one proof - the structure
 * of arecibo's spartan::batched::BatchedRelaxedR1CSSNARK, SuperNova's compressor (/root/reference/src/proof/supernova.rs:110, 293-302): one
 * outer and one inner sum-check shared through powers of a challenge, all 2 n evaluation claims batched to one point, one opening.  The
 * protocol is the repository's own (oracle/spartan_fast.py: prove_batched / verify_batched), as for the single-instance call.  Per
 * instance the arguments of lurk_hip_spartan_prove_dev.  With ell_x = log2(max num_cons), ell_y = log2(max num_vars) + 1,
 * N = max(num_cons, num_vars) over the instances, the outputs (host, canonical) are: polys_outer ell_x x 4 x 32 B, claims_outer
 * n x 3 x 32, evals_e n x 32, polys_inner ell_y x 3 x 32, evals_w n x 32, polys_batch log2(N) x 3 x 32, evals_batch 2 n x 32 (W_i, E_i
 * per instance), ipa_l / ipa_r log2(N) x 96 (Jacobians), ipa_a 32. */
typedef struct lurk_hip_spartan_instance {
    const lurk_hip_r1cs* shape;
    const lurk_hip_r1cs* shape_t;
    size_t num_cons, num_vars, num_io;
    const void* x32_canonical;
    const void* u32_canonical;
    const void* d_w32_mont;
    const void* d_e32_mont;
    const void* comm_w_jacobian96;
    const void* comm_e_jacobian96;
} lurk_hip_spartan_instance;
typedef struct lurk_hip_spartan_batch_proof {
    void* polys_outer;
    void* claims_outer;
    void* evals_e;
    void* polys_inner;
    void* evals_w;
    void* polys_batch;
    void* evals_batch;
    void* ipa_l;
    void* ipa_r;
    void* ipa_a;
} lurk_hip_spartan_batch_proof;
int lurk_hip_spartan_prove_batch_dev(const lurk_hip_spartan_instance* instances, size_t n_instances, lurk_hip_msm_ctx* key,
                                     const void* ck_c_jacobian96, const void* label, size_t label_len, lurk_hip_spartan_batch_proof* out,
                                     void* stream);

/* ---- synthetic inputs (bench / tests) -------------------------------------
 * SplitMix64 counter mode, seed 0x4C55524B.  dist 0 = uniform, 1 = witness-like. */
int lurk_hip_synth_scalars_dev(int field_id, uint64_t stream_id, int dist, size_t first, size_t n,
                               void* d_out32, int out_mont, void* stream);
/* bases P_i = [k_i]G, k_i = uniform(stream 0, first+i) (0 -> 1); affine Montgomery, 64 B each */
int lurk_hip_synth_bases_dev(int curve, size_t first, size_t n, void* d_out_affine64, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LURK_HIP_H */

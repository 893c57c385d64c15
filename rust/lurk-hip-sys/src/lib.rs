//! Bindings of `include/lurk_hip.h` (module [`ffi`], generated) and the safe wrappers a lurk-beta / arecibo build would use:
//!
//! * [`pasta_msm`]: the API of the `pasta-msm` crate (`pallas(points, scalars)`, `vesta(..)`, and the resident-key form
//!   `init` / `with` of the argumentcomputer fork) over `mult_pippenger_*` / `lurk_hip_msm_ctx_*` - what arecibo's
//!   `DlogGroup::vartime_multiscalar_mul` calls from `CommitmentEngine::commit`
//!   (callers in lurk-beta: `/root/reference/src/proof/nova.rs:287-293`, `supernova.rs:231-244`);
//! * [`FoldingContext`]: one curve's half of `RecursiveSNARK::prove_step` (`NIFS::prove`) with the running pair resident in HBM;
//! * [`poseidon`]: batched `PoseidonCache::hash{3,4,6,8}` (`/root/reference/src/hash.rs:180-204`);
//! * [`store`]: `HipPoseidonCache` (the shape of `PoseidonCache<F>`), `HipStoreHasher` (the shape of `StoreHasher`,
//!   `/root/reference/src/lem/store_core.rs:10-14`, layouts of `/root/reference/src/lem/store.rs:29-78`) and `hydrate`
//!   (`hydrate_z_cache`, `store_core.rs:256-269`, as one level-batched call);
//! * [`params`]: the run-time parameter blocks of the two restatements written from memory - Nova's random oracle
//!   (`lurk_hip_ro_params`) and `from_label` (`lurk_hip_ck_params`) - so that the first run beside arecibo can move one field at a time
//!   without rebuilding the library (callers: `/root/reference/src/proof/nova.rs:196-216, 282-295`);
//! * [`dump`]: LURKDUMP writers - `R1CSShape`, the fresh witnesses of consecutive steps and the `CommitmentKey` as the files
//!   `bench.py --workload fold_step --shape-file .. --witness-file ..` measures (`/root/reference/benches/fibonacci.rs:98-122`).
//!
//! Never compiled in the container this repository is built in (no Rust toolchain); the extern block cannot drift from the header
//! (see `rust/gen_sys.py`), the wrappers are a reading aid for the maintainer who wires the feature in.
pub mod dump;
pub mod ffi;
pub mod params;
pub mod store;
pub use ffi::*;

use core::ffi::{c_int, c_void};
use std::ffi::CStr;

/// A non-zero return code of the library with `lurk_hip_last_error()`'s message.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: c_int,
    pub message: String,
}
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "lurk_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

pub fn check(rc: c_int) -> Result<(), Error> {
    if rc == LURK_HIP_OK {
        return Ok(());
    }
    // SAFETY: the library returns a pointer to a thread-local, NUL-terminated buffer that lives until the next call on this thread
    let message = unsafe { CStr::from_ptr(lurk_hip_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code: rc, message })
}

/// The `pasta-msm` crate's surface.  `pasta_curves` with `repr-c` lays `Affine` out as 64 bytes and `Point` as 96 bytes of
/// Montgomery limbs, which is what the C ABI reads and writes; scalars are handed over in Montgomery form (`is_mont = true`),
/// as pasta-msm does.
pub mod pasta_msm {
    use super::*;
    use pasta_curves::{pallas, vesta};

    macro_rules! curve_mod {
        ($m:ident, $curve_id:expr, $oneshot:ident, $ctx:ident) => {
            /// `pasta_msm::$m(points, scalars)`: everything in host memory, result on return (aborts with the library's message on
            /// failure, as the C symbol of pasta-msm does).
            pub fn $m(points: &[$m::Affine], scalars: &[$m::Scalar]) -> $m::Point {
                let n = points.len().min(scalars.len());
                let mut out = $m::Point::default();
                #[cfg(feature = "oneshot-key-cache")]
                unsafe {
                    lurk_hip_msm_oneshot_key_cache(1);
                }
                // SAFETY: both slices hold at least n elements of the layouts named above; out is 96 writable bytes
                unsafe { $oneshot((&mut out as *mut $m::Point).cast(), points.as_ptr().cast(), n, scalars.as_ptr().cast(), true) };
                out
            }

            /// A commitment key resident in HBM (`CommitmentKey` is constant for a whole proof, `/root/reference/src/proof/nova.rs:196-216`):
            /// `init(points)` / `with(&ctx, scalars)` of the argumentcomputer pasta-msm fork.
            pub struct $ctx(*mut lurk_hip_msm_ctx);
            // SAFETY: every entry point of the library is thread-safe and a handle records its device
            unsafe impl Send for $ctx {}
            unsafe impl Sync for $ctx {}
            impl $ctx {
                pub fn init(points: &[$m::Affine], precompute: bool) -> Result<Self, Error> {
                    let mut p = core::ptr::null_mut();
                    let flags = if precompute { LURK_MSM_FLAG_PRECOMPUTE } else { 0 };
                    check(unsafe { lurk_hip_msm_ctx_create(&mut p, $curve_id, points.as_ptr().cast(), points.len(), flags) })?;
                    Ok(Self(p))
                }
                /// `CE::commit(ck, v)`: MSM over `ck[..v.len()]`
                pub fn with(&self, scalars: &[$m::Scalar]) -> Result<$m::Point, Error> {
                    let mut out = $m::Point::default();
                    check(unsafe { lurk_hip_msm_ctx_run(self.0, (&mut out as *mut $m::Point).cast(), scalars.as_ptr().cast(), scalars.len(), 1) })?;
                    Ok(out)
                }
                /// Scalars already in device memory (e.g. a witness assembled by `lurk_hip_frames_witness_dev`); up to
                /// `LURK_MSM_SLOTS` commitments in flight: `submit` then `wait` on the same slot.
                ///
                /// # Safety
                /// `d_scalars` must point to `n` 32-byte Montgomery scalars in device memory that stay valid until `wait`.
                pub unsafe fn submit(&self, slot: c_int, d_scalars: *const c_void, n: usize, stream: *mut c_void, mode: c_int) -> Result<(), Error> {
                    check(lurk_hip_msm_ctx_submit_dev_mode(self.0, slot, d_scalars, n, 1, stream, mode))
                }
                pub fn wait(&self, slot: c_int) -> Result<$m::Point, Error> {
                    let mut out = $m::Point::default();
                    check(unsafe { lurk_hip_msm_ctx_wait(self.0, slot, (&mut out as *mut $m::Point).cast()) })?;
                    Ok(out)
                }
                /// The key folded by the weights of k inner-product rounds at once (`ck_k[p] = sum_b w_b * ck[b m + p]`, arecibo `ipa_pc`):
                /// `weights` are Montgomery scalars (a power of two of them, at most 2^12), the `n / weights.len()` affine points land in
                /// device memory.  Window-table keys only; `lurk_hip_ipa_prove_dev` calls this by itself.
                ///
                /// # Safety
                /// `d_out` must point to `n / weights.len()` 64-byte records of device memory.
                pub unsafe fn fold_key(&self, n: usize, weights: &[$m::Scalar], d_out: *mut c_void, stream: *mut c_void) -> Result<(), Error> {
                    check(lurk_hip_msm_ctx_fold_key_dev(self.0, n, weights.as_ptr().cast(), weights.len(), d_out, stream))
                }
                pub fn as_ptr(&self) -> *mut lurk_hip_msm_ctx {
                    self.0
                }
            }
            impl Drop for $ctx {
                fn drop(&mut self) {
                    unsafe { lurk_hip_msm_ctx_destroy(self.0) };
                }
            }
        };
    }
    curve_mod!(pallas, LURK_CURVE_PALLAS, mult_pippenger_pallas, MSMContextPallas);
    curve_mod!(vesta, LURK_CURVE_VESTA, mult_pippenger_vesta, MSMContextVesta);
}

/// An R1CS shape resident in HBM (arecibo `R1CSShape { A, B, C }` in CSR form, coefficients in Montgomery form).
pub struct R1csShape(*mut lurk_hip_r1cs);
unsafe impl Send for R1csShape {}
unsafe impl Sync for R1csShape {}
/// One sparse matrix as arecibo's `SparseMatrix { data, indices, indptr }` holds it.
pub struct Csr<'a> {
    pub indptr: &'a [u64],
    pub indices: &'a [u64],
    /// `indices.len()` field elements of 32 bytes, Montgomery form
    pub data: &'a [u8],
}
impl R1csShape {
    pub fn new(field_id: c_int, num_cons: usize, num_vars: usize, num_io: usize, a: Csr, b: Csr, c: Csr) -> Result<Self, Error> {
        let mut p = core::ptr::null_mut();
        check(unsafe {
            lurk_hip_r1cs_create(&mut p, field_id, num_cons, num_vars, num_io, a.indptr.as_ptr(), a.indices.as_ptr(), a.data.as_ptr().cast(),
                                 b.indptr.as_ptr(), b.indices.as_ptr(), b.data.as_ptr().cast(), c.indptr.as_ptr(), c.indices.as_ptr(),
                                 c.data.as_ptr().cast())
        })?;
        Ok(Self(p))
    }
    pub fn as_ptr(&self) -> *mut lurk_hip_r1cs {
        self.0
    }
}
impl Drop for R1csShape {
    fn drop(&mut self) {
        unsafe { lurk_hip_r1cs_destroy(self.0) };
    }
}

/// The single-instance compressing prover as one call (`lurk_hip_spartan_prove_dev`): sum-checks, claim batching and the opening with the
/// Keccak transcript inside.  The protocol is the library's own, not arecibo's byte for byte.
pub struct SpartanProof {
    pub polys_outer: Vec<u8>,  // log2(num_cons) x 4 x 32 B, canonical
    pub claims_outer: [u8; 96],
    pub eval_e: [u8; 32],
    pub polys_inner: Vec<u8>,  // (log2(num_vars) + 1) x 3 x 32 B
    pub eval_w: [u8; 32],
    pub polys_batch: Vec<u8>,  // log2(N) x 3 x 32 B, N = max(num_cons, num_vars)
    pub evals_batch: [u8; 64],
    pub ipa_l: Vec<u8>,        // log2(N) x 96 B Jacobians
    pub ipa_r: Vec<u8>,
    pub ipa_a: [u8; 32],
}
/// # Safety
/// `d_w` / `d_e`: `num_vars` / `num_cons` Montgomery scalars in device memory; `shape_t` is the transpose of `shape` (2 `num_vars` rows over
/// `num_cons` columns); `key` holds at least max(`num_cons`, `num_vars`) points and committed W and E.
#[allow(clippy::too_many_arguments)]
pub unsafe fn spartan_prove(shape: &R1csShape, shape_t: &R1csShape, num_cons: usize, num_vars: usize, key: *mut lurk_hip_msm_ctx, ck_c: &[u8; 96], x: &[u8], u: &[u8; 32],
                            d_w: *const c_void, d_e: *const c_void, comm_w: &[u8; 96], comm_e: &[u8; 96], label: &[u8], stream: *mut c_void) -> Result<SpartanProof, Error> {
    let log2 = |n: usize| n.trailing_zeros() as usize;
    let (ell_x, ell_y, ell) = (log2(num_cons), log2(num_vars) + 1, log2(num_cons.max(num_vars)));
    let mut p = SpartanProof {
        polys_outer: vec![0; ell_x * 128], claims_outer: [0; 96], eval_e: [0; 32], polys_inner: vec![0; ell_y * 96], eval_w: [0; 32],
        polys_batch: vec![0; ell.max(1) * 96], evals_batch: [0; 64], ipa_l: vec![0; ell.max(1) * 96], ipa_r: vec![0; ell.max(1) * 96], ipa_a: [0; 32],
    };
    let mut out = lurk_hip_spartan_proof {
        polys_outer: p.polys_outer.as_mut_ptr().cast(), claims_outer: p.claims_outer.as_mut_ptr().cast(), eval_e: p.eval_e.as_mut_ptr().cast(),
        polys_inner: p.polys_inner.as_mut_ptr().cast(), eval_w: p.eval_w.as_mut_ptr().cast(), polys_batch: p.polys_batch.as_mut_ptr().cast(),
        evals_batch: p.evals_batch.as_mut_ptr().cast(), ipa_l: p.ipa_l.as_mut_ptr().cast(), ipa_r: p.ipa_r.as_mut_ptr().cast(), ipa_a: p.ipa_a.as_mut_ptr().cast(),
    };
    check(lurk_hip_spartan_prove_dev(shape.as_ptr(), shape_t.as_ptr(), num_cons, num_vars, x.len() / 32, key, ck_c.as_ptr().cast(), x.as_ptr().cast(), u.as_ptr().cast(),
                                     d_w, d_e, comm_w.as_ptr().cast(), comm_e.as_ptr().cast(), label.as_ptr().cast(), label.len(), &mut out, stream))?;
    Ok(p)
}

/// The batched compressing prover as one call (`lurk_hip_spartan_prove_batch_dev`): n relaxed instances of different shapes under one key,
/// one proof - the structure of arecibo's `BatchedRelaxedR1CSSNARK`, SuperNova's compressor (`/root/reference/src/proof/supernova.rs:110,
/// 293-302`).  Sizes of the outputs: `include/lurk_hip.h`.
pub struct SpartanBatchProof {
    pub polys_outer: Vec<u8>,   // log2(max num_cons) x 4 x 32 B, canonical
    pub claims_outer: Vec<u8>,  // n x 3 x 32 B
    pub evals_e: Vec<u8>,       // n x 32 B
    pub polys_inner: Vec<u8>,   // (log2(max num_vars) + 1) x 3 x 32 B
    pub evals_w: Vec<u8>,       // n x 32 B
    pub polys_batch: Vec<u8>,   // log2(N) x 3 x 32 B
    pub evals_batch: Vec<u8>,   // 2 n x 32 B
    pub ipa_l: Vec<u8>,         // log2(N) x 96 B Jacobians
    pub ipa_r: Vec<u8>,
    pub ipa_a: [u8; 32],
}
/// # Safety
/// Every instance's pointers obey the contract of [`spartan_prove`]; `key` holds at least max(`num_cons`, `num_vars`) points over all instances.
pub unsafe fn spartan_prove_batch(instances: &[lurk_hip_spartan_instance], key: *mut lurk_hip_msm_ctx, ck_c: &[u8; 96], label: &[u8], stream: *mut c_void)
                                  -> Result<SpartanBatchProof, Error> {
    let log2 = |n: usize| n.trailing_zeros() as usize;
    let n = instances.len();
    let max_nc = instances.iter().map(|i| i.num_cons).max().unwrap_or(2);
    let max_nv = instances.iter().map(|i| i.num_vars).max().unwrap_or(2);
    let (ell_x, ell_y, ell) = (log2(max_nc), log2(max_nv) + 1, log2(max_nc.max(max_nv)));
    let mut p = SpartanBatchProof {
        polys_outer: vec![0; ell_x * 128], claims_outer: vec![0; n * 96], evals_e: vec![0; n * 32], polys_inner: vec![0; ell_y * 96], evals_w: vec![0; n * 32],
        polys_batch: vec![0; ell.max(1) * 96], evals_batch: vec![0; 2 * n * 32], ipa_l: vec![0; ell.max(1) * 96], ipa_r: vec![0; ell.max(1) * 96], ipa_a: [0; 32],
    };
    let mut out = lurk_hip_spartan_batch_proof {
        polys_outer: p.polys_outer.as_mut_ptr().cast(), claims_outer: p.claims_outer.as_mut_ptr().cast(), evals_e: p.evals_e.as_mut_ptr().cast(),
        polys_inner: p.polys_inner.as_mut_ptr().cast(), evals_w: p.evals_w.as_mut_ptr().cast(), polys_batch: p.polys_batch.as_mut_ptr().cast(),
        evals_batch: p.evals_batch.as_mut_ptr().cast(), ipa_l: p.ipa_l.as_mut_ptr().cast(), ipa_r: p.ipa_r.as_mut_ptr().cast(), ipa_a: p.ipa_a.as_mut_ptr().cast(),
    };
    check(lurk_hip_spartan_prove_batch_dev(instances.as_ptr(), n, key, ck_c.as_ptr().cast(), label.as_ptr().cast(), label.len(), &mut out, stream))?;
    Ok(p)
}

/// One curve's half of `RecursiveSNARK::prove_step`: the running relaxed pair (z1 = [W | u | X], E) and the running instance stay in
/// the context; `step` is `NIFS::prove` (commit W2, cross term, commit T, r from the transcript, fold).
/// Borrowing the shape and the key ties their lifetimes to the context's, as the C ABI requires.
pub struct FoldingContext<'a> {
    h: *mut lurk_hip_fold_ctx,
    num_io: usize,
    _shape: core::marker::PhantomData<&'a R1csShape>,
}
/// What a step returns: comm_W2, comm_T (96-byte Jacobians) and the challenge r (32 bytes, Montgomery).
pub struct StepOutput {
    pub comm_w2: [u8; 96],
    pub comm_t: [u8; 96],
    pub r: [u8; 32],
}
impl<'a> FoldingContext<'a> {
    pub fn new(curve: c_int, shape: &'a R1csShape, key: *mut lurk_hip_msm_ctx, num_io: usize) -> Result<Self, Error> {
        let mut h = core::ptr::null_mut();
        check(unsafe { lurk_hip_fold_ctx_create(&mut h, curve, shape.as_ptr(), key) })?;
        Ok(Self { h, num_io, _shape: core::marker::PhantomData })
    }
    /// `w2`: the fresh witness (host memory, Montgomery); `x2`: its public IO; `pp_digest`: 32 canonical bytes.
    pub fn step(&mut self, w2: &[u8], x2: &[u8], pp_digest: &[u8; 32]) -> Result<StepOutput, Error> {
        assert_eq!(x2.len(), 32 * self.num_io);
        let mut out = StepOutput { comm_w2: [0; 96], comm_t: [0; 96], r: [0; 32] };
        check(unsafe {
            lurk_hip_fold_step(self.h, w2.as_ptr().cast(), 0, core::ptr::null_mut(), x2.as_ptr().cast(), pp_digest.as_ptr().cast(),
                               out.comm_w2.as_mut_ptr().cast(), out.comm_t.as_mut_ptr().cast(), out.r.as_mut_ptr().cast())
        })?;
        Ok(out)
    }
    /// The two halves for a caller with its own transcript: `begin` returns (comm_W2, comm_T), `finish(r)` folds.
    pub fn begin(&mut self, w2: &[u8], x2: &[u8]) -> Result<([u8; 96], [u8; 96]), Error> {
        let (mut cw, mut ct) = ([0u8; 96], [0u8; 96]);
        check(unsafe {
            lurk_hip_fold_step_begin(self.h, w2.as_ptr().cast(), 0, core::ptr::null_mut(), x2.as_ptr().cast(), cw.as_mut_ptr().cast(),
                                     ct.as_mut_ptr().cast())
        })?;
        Ok((cw, ct))
    }
    pub fn finish(&mut self, r_mont: &[u8; 32]) -> Result<(), Error> {
        check(unsafe { lurk_hip_fold_step_finish(self.h, r_mont.as_ptr().cast()) })
    }
    /// The halves with the library's transcript: set the digest of the public parameters once; every `begin` then absorbs pp_digest and
    /// the running instance while the device works, and `challenge` finishes behind comm_T with one permutation (r in Montgomery form).
    pub fn set_pp_digest(&mut self, pp_digest: &[u8; 32]) -> Result<(), Error> {
        check(unsafe { lurk_hip_fold_ctx_set_pp_digest(self.h, pp_digest.as_ptr().cast()) })
    }
    pub fn challenge(&mut self) -> Result<[u8; 32], Error> {
        let mut r = [0u8; 32];
        check(unsafe { lurk_hip_fold_step_challenge(self.h, r.as_mut_ptr().cast()) })?;
        Ok(r)
    }
    /// A hook into every step: called once per `step` / `begin` / `begin_prefetched`, on the calling thread, after the step's device work
    /// has been enqueued and before the call blocks on the commitments - where the next witness's slot traces are enqueued
    /// (`lurk_hip_slot_witness_dev`), so that they run beside the step.  A non-zero return fails the step (it is rolled back).
    ///
    /// # Safety
    /// `hook` and `user` must stay valid until the hook is replaced, removed (`None`) or the context dropped; the hook must not call
    /// into this context.
    pub unsafe fn set_submit_hook(&mut self, hook: lurk_hip_fold_submit_hook_fn, user: *mut c_void) -> Result<(), Error> {
        check(lurk_hip_fold_ctx_set_submit_hook(self.h, hook, user))
    }
    /// Staging ahead across devices: `helper` is the same commitment key resident on another device; instances staged with
    /// `prefetch` are committed on the helpers in turn while this context's device folds.
    ///
    /// # Safety
    /// `helper` must outlive the context.
    pub unsafe fn add_helper(&mut self, helper: *mut lurk_hip_msm_ctx) -> Result<(), Error> {
        check(lurk_hip_fold_ctx_add_helper(self.h, helper))
    }
    /// Stage positions `[offset, offset + range.len() / 32)` of the next fresh witness (host memory, Montgomery) and start its commitment.
    pub fn prefetch(&mut self, range: &[u8], offset: usize) -> Result<(), Error> {
        check(unsafe { lurk_hip_fold_step_prefetch(self.h, range.as_ptr().cast(), offset, range.len() / 32, 0, core::ptr::null_mut()) })
    }
    /// Open the step of the oldest staged instance; `patches`: the ranges of W2 known only now.
    pub fn begin_prefetched(&mut self, patches: &[lurk_hip_w2_patch], x2: &[u8]) -> Result<([u8; 96], [u8; 96]), Error> {
        let (mut cw, mut ct) = ([0u8; 96], [0u8; 96]);
        check(unsafe {
            lurk_hip_fold_step_begin_prefetched(self.h, patches.as_ptr(), patches.len(), x2.as_ptr().cast(), cw.as_mut_ptr().cast(), ct.as_mut_ptr().cast())
        })?;
        Ok((cw, ct))
    }
}
impl Drop for FoldingContext<'_> {
    fn drop(&mut self) {
        unsafe { lurk_hip_fold_ctx_destroy(self.h) };
    }
}

/// Batched Poseidon behind `PoseidonCache::hash{3,4,6,8}` (`/root/reference/src/hash.rs:180-204`): `preimages` holds `n * arity`
/// canonical 32-byte elements (`to_repr()`), the result `n` digests.  `arity` must be 3, 4, 6 or 8 (`hash.rs:19-29`).
pub mod poseidon {
    use super::*;
    pub fn hash_batch(field_id: c_int, arity: usize, preimages: &[u8]) -> Result<Vec<u8>, Error> {
        assert!(matches!(arity, 3 | 4 | 6 | 8), "unsupported arity");
        assert_eq!(preimages.len() % (32 * arity), 0);
        let n = preimages.len() / (32 * arity);
        let mut out = vec![0u8; 32 * n];
        check(unsafe { lurk_hip_poseidon_batch(field_id, arity as c_int, preimages.as_ptr().cast(), n, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
}

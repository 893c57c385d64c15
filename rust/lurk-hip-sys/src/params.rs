//! Run-time parameters of the two restatements that decide interoperability with arecibo (`include/lurk_hip.h`, round 6):
//! [`RoParams`] = `lurk_hip_ro_params` (the Nova random oracle behind `NIFS::prove`, `/root/reference/src/proof/nova.rs:282-295`) and
//! [`CkParams`] = `lurk_hip_ck_params` (`CommitmentKey::setup` = `from_label`, `/root/reference/src/proof/nova.rs:196-216`).
//! Both were written from memory of un-vendored crates (`/root/reference/Cargo.toml:127-131`); every recalled constant is a field,
//! process-wide, with the recalled value as its default.  The intended first run inside arecibo:
//!
//! ```ignore
//! // 1. write ONE probe record from a real step (dump::ProbeWriter) and read the verdict:
//! //        python -m lurk_beta_amd.dump probe step0.lurkdump --search
//! // 2. apply what the search printed, e.g. {"ro": {"point_elements": 2}}:
//! let mut ro = params::RoParams::get()?;
//! ro.0.point_elements = 2;
//! ro.set()?;
//! ```
//! Never compiled in the container this repository is built in (no Rust toolchain).
use crate::{check, ffi, Error};

/// `lurk_hip_ro_params`: arity, IO-pattern tag, absorb order of `(pp_digest, U1, U2, comm_T)`, the order inside a relaxed / fresh
/// instance, commitment encoding, limb split of `X`, squeeze width.
#[derive(Clone, Copy, Debug)]
pub struct RoParams(pub ffi::lurk_hip_ro_params);

impl RoParams {
    /// The block in force.
    pub fn get() -> Result<Self, Error> {
        let mut p = unsafe { std::mem::zeroed::<ffi::lurk_hip_ro_params>() };
        check(unsafe { ffi::lurk_hip_ro_params_get(&mut p) })?;
        Ok(Self(p))
    }
    /// Install this block (validated by the library; a refused block changes nothing).
    pub fn set(&self) -> Result<(), Error> {
        let mut p = self.0;
        p.struct_size = std::mem::size_of::<ffi::lurk_hip_ro_params>() as u32;
        check(unsafe { ffi::lurk_hip_ro_params_set(&p) })
    }
    /// Back to the defaults (the values of rounds 1-5).
    pub fn reset() -> Result<(), Error> {
        check(unsafe { ffi::lurk_hip_ro_params_set(std::ptr::null()) })
    }
}

/// `lurk_hip_ck_params`: the XOF over the label, bytes per point, and the three parts of `hash_to_curve`'s domain-separation tag.
#[derive(Clone, Copy, Debug)]
pub struct CkParams(pub ffi::lurk_hip_ck_params);

impl CkParams {
    pub fn get() -> Result<Self, Error> {
        let mut p = unsafe { std::mem::zeroed::<ffi::lurk_hip_ck_params>() };
        check(unsafe { ffi::lurk_hip_ck_params_get(&mut p) })?;
        Ok(Self(p))
    }
    pub fn set(&self) -> Result<(), Error> {
        let mut p = self.0;
        p.struct_size = std::mem::size_of::<ffi::lurk_hip_ck_params>() as u32;
        check(unsafe { ffi::lurk_hip_ck_params_set(&p) })
    }
    pub fn reset() -> Result<(), Error> {
        check(unsafe { ffi::lurk_hip_ck_params_set(std::ptr::null()) })
    }
    /// `domain_prefix` from a Rust string (at most 31 bytes; NUL-terminated in place).
    pub fn set_domain_prefix(&mut self, prefix: &str) {
        assert!(prefix.len() < self.0.domain_prefix.len());
        self.0.domain_prefix = [0; 32];
        for (d, b) in self.0.domain_prefix.iter_mut().zip(prefix.bytes()) {
            *d = b as core::ffi::c_char;
        }
    }
}

/// The first `n` points of `from_label(label)` mapped on the HOST (64-byte affine Montgomery records; no device): what a probe
/// record's key points are compared with.
pub fn ck_from_label_host(curve: core::ffi::c_int, label: &[u8], n: usize) -> Result<Vec<[u8; 64]>, Error> {
    let mut out = vec![[0u8; 64]; n];
    check(unsafe { ffi::lurk_hip_ck_from_label_host(curve, label.as_ptr().cast(), label.len(), n, out.as_mut_ptr().cast()) })?;
    Ok(out)
}

//! The two Poseidon hooks lurk-beta's store has, over the library:
//!
//! * [`HipPoseidonCache`] - the shape of `PoseidonCache<F>` (`/root/reference/src/hash.rs:86-114, 180-204`): `hash3` / `hash4` /
//!   `hash6` / `hash8` of ONE preimage with the same memoisation, a miss hashed by the library's HOST Poseidon
//!   (`lurk_hip_poseidon_hash_host`: ~20 us, no launch), and `hash_many` for whoever holds a batch (`lurk_hip_poseidon_batch`);
//! * [`HipStoreHasher`] - the shape of `StoreHasher<Tag, FWrap<F>>` (`/root/reference/src/lem/store_core.rs:10-14`) with the preimage
//!   layouts of `impl StoreHasher for PoseidonCache` (`/root/reference/src/lem/store.rs:29-78`): `hash_ptrs` (2, 3 or 4 tagged
//!   pointers -> hash4 / hash6 / hash8), `hash_compact`, `hash_commitment`;
//! * [`hydrate`] - `StoreCore::hydrate_z_cache` (`/root/reference/src/lem/store_core.rs:256-269`) as ONE call: the dehydrated DAG in
//!   topological order, every level hashed as a batch (`lurk_hip_store_hydrate`: wide levels on the device, narrow ones on the host).
//!
//! Field elements cross as their canonical 32 bytes (`PrimeField::to_repr()`); a tag is `F::from(tag as u64)` (`src/tag.rs:99-101`).
//! A generic `impl<F: LurkField> StoreHasher<Tag, FWrap<F>> for HipStoreHasher` is four lines of `to_repr` / `from_repr` around these
//! methods and lives in lurk-beta (it needs its `Tag` and `FWrap` types).  Never compiled in the container this repository is built in.
use super::*;
use std::collections::HashMap;
use std::sync::Mutex;

pub type Digest = [u8; 32];

/// `F::from(tag as u64).to_repr()`
pub fn tag_to_field(tag: u16) -> Digest {
    let mut d = [0u8; 32];
    d[..2].copy_from_slice(&tag.to_le_bytes());
    d
}

/// `PoseidonCache<F>` over the library; `field_id` is one of `LURK_FIELD_*`.
pub struct HipPoseidonCache {
    pub field_id: c_int,
    memo: Mutex<HashMap<Vec<u8>, Digest>>,
}
impl HipPoseidonCache {
    pub fn new(field_id: c_int) -> Self {
        Self { field_id, memo: Mutex::new(HashMap::new()) }
    }
    fn hash(&self, arity: usize, preimage: &[Digest]) -> Result<Digest, Error> {
        assert_eq!(preimage.len(), arity);
        let key: Vec<u8> = preimage.concat();
        if let Some(d) = self.memo.lock().unwrap().get(&key) {
            return Ok(*d);
        }
        let mut out = [0u8; 32];
        check(unsafe { lurk_hip_poseidon_hash_host(self.field_id, arity as c_int, key.as_ptr().cast(), 1, out.as_mut_ptr().cast()) })?;
        self.memo.lock().unwrap().insert(key, out);
        Ok(out)
    }
    pub fn hash3(&self, p: &[Digest; 3]) -> Result<Digest, Error> { self.hash(3, p) }
    pub fn hash4(&self, p: &[Digest; 4]) -> Result<Digest, Error> { self.hash(4, p) }
    pub fn hash6(&self, p: &[Digest; 6]) -> Result<Digest, Error> { self.hash(6, p) }
    pub fn hash8(&self, p: &[Digest; 8]) -> Result<Digest, Error> { self.hash(8, p) }
    /// n preimages of one arity at once (device): what a caller that already holds a level of the DAG should use
    pub fn hash_many(&self, arity: usize, preimages: &[Digest]) -> Result<Vec<Digest>, Error> {
        assert!(matches!(arity, 3 | 4 | 6 | 8) && preimages.len() % arity == 0);
        let n = preimages.len() / arity;
        let mut out = vec![[0u8; 32]; n];
        check(unsafe { lurk_hip_poseidon_batch(self.field_id, arity as c_int, preimages.as_ptr().cast(), n, out.as_mut_ptr().cast()) })?;
        Ok(out)
    }
}

/// `StoreHasher<Tag, FWrap<F>>` (`store_core.rs:10-14`) with `store.rs:29-78`'s layouts; a tagged pointer is `(tag, digest)`.
pub struct HipStoreHasher {
    pub cache: HipPoseidonCache,
}
impl HipStoreHasher {
    pub fn new(field_id: c_int) -> Self {
        Self { cache: HipPoseidonCache::new(field_id) }
    }
    /// 2 pointers -> hash4(tag_a, h_a, tag_b, h_b); 3 -> hash6; 4 -> hash8 (`store.rs:30-66`)
    pub fn hash_ptrs(&self, ptrs: &[(u16, Digest)]) -> Result<Digest, Error> {
        let pre: Vec<Digest> = ptrs.iter().flat_map(|(t, h)| [tag_to_field(*t), *h]).collect();
        match ptrs.len() {
            2 => self.cache.hash(4, &pre),
            3 => self.cache.hash(6, &pre),
            4 => self.cache.hash(8, &pre),
            _ => unimplemented!("hash_ptrs takes 2, 3 or 4 pointers (store.rs:66)"),
        }
    }
    /// hash4(d1, t2, d2, d3) (`store.rs:75-77`)
    pub fn hash_compact(&self, d1: Digest, t2: u16, d2: Digest, d3: Digest) -> Result<Digest, Error> {
        self.cache.hash4(&[d1, tag_to_field(t2), d2, d3])
    }
    /// hash3(secret, tag, hash) (`store.rs:70-73`)
    pub fn hash_commitment(&self, secret: Digest, payload: (u16, Digest)) -> Result<Digest, Error> {
        self.cache.hash3(&[secret, tag_to_field(payload.0), payload.1])
    }
}

/// `hydrate_z_cache` in one call: `nodes` in topological order (children first; `kind` = `LURK_NODE_*`), `values` the atoms' and
/// secrets' field elements; returns the digest of every node and the depth of the DAG.
pub fn hydrate(field_id: c_int, nodes: &[lurk_hip_store_node], values: &[Digest]) -> Result<(Vec<Digest>, usize), Error> {
    let mut out = vec![[0u8; 32]; nodes.len()];
    let mut levels = 0usize;
    check(unsafe {
        lurk_hip_store_hydrate(field_id, nodes.as_ptr(), nodes.len(), values.as_ptr().cast(), values.len(), out.as_mut_ptr().cast(), &mut levels)
    })?;
    Ok((out, levels))
}

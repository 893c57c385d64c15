//! LURKDUMP writers: the R1CS shape, the fresh witnesses and the commitment key of a running lurk-beta prover as the three files
//! `bench.py --workload fold_step --shape-file .. --witness-file .. [--key-file ..]` reads (format and reader:
//! `lurk_beta_amd/dump.py`; a CPU test keeps the constants below equal to the reader's).
//!
//! Everything bulk is a byte copy: `pasta_curves` with `repr-c` holds a field element as four little-endian `u64` Montgomery limbs
//! and an affine point as `{x, y}`, arecibo's `SparseMatrix` holds `indptr` / `indices` as `Vec<usize>` - so `ENC_MONTGOMERY` files
//! are written with no per-element conversion (a canonical dump, `ENC_CANONICAL`, is `to_repr()` per element).
//!
//! Where the pieces come from in lurk-beta (`/root/reference/benches/fibonacci.rs:98-122`, `src/proof/nova.rs:196-216, 282-295`,
//! `src/lem/multiframe.rs:856-920`) - inside arecibo, where the fields are visible:
//!
//! ```ignore
//! let s = &pp.pp.circuit_shape_primary.r1cs_shape;                 // R1CSShape<E1>
//! dump::shape(path_s, FIELD_PALLAS_FQ, ENC_MONTGOMERY, s.num_cons, s.num_vars, s.num_io,
//!             [(&s.A.indptr, &s.A.indices, &s.A.data), (&s.B.indptr, &s.B.indices, &s.B.data), (&s.C.indptr, &s.C.indices, &s.C.data)])?;
//! dump::key(path_k, CURVE_PALLAS, ENC_MONTGOMERY, &pp.pp.ck_primary.ck)?;       // Vec<pallas::Affine>
//! // in RecursiveSNARK::prove_step, after the step circuit is synthesized (one call per folding step):
//! writer.step(&l_w_primary.W, &l_u_primary.X)?;                   // R1CSWitness::W, R1CSInstance::X
//! ```
//! Never compiled in the container this repository is built in (no Rust toolchain).
use std::fs::File;
use std::io::{BufWriter, Result, Seek, SeekFrom, Write};
use std::mem::size_of;
use std::path::Path;

pub const MAGIC: &[u8; 8] = b"LURKDUMP";
pub const VERSION: u32 = 1;
pub const KIND_SHAPE: u32 = 1;
pub const KIND_WITNESS: u32 = 2;
pub const KIND_KEY: u32 = 3;
pub const KIND_PROBE: u32 = 4;
pub const ENC_CANONICAL: u32 = 0;
pub const ENC_MONTGOMERY: u32 = 1;
pub const HEADER_BYTES: usize = 64;

fn header(w: &mut impl Write, kind: u32, id: u32, encoding: u32, a: u64, b: u64, c: u64) -> Result<()> {
    w.write_all(MAGIC)?;
    for v in [VERSION, kind, id, encoding] {
        w.write_all(&v.to_le_bytes())?;
    }
    for v in [a, b, c] {
        w.write_all(&v.to_le_bytes())?;
    }
    w.write_all(&[0u8; 16])
}

/// The bytes of a slice of plain-old-data values (field elements of 32 bytes, affine points of 64, `usize` indices of 8).
fn bytes_of<T>(v: &[T]) -> &[u8] {
    // SAFETY: T is one of the repr(C) limb arrays / integers named above: no padding, every bit pattern readable
    unsafe { std::slice::from_raw_parts(v.as_ptr().cast::<u8>(), v.len() * size_of::<T>()) }
}

/// `R1CSShape`: three CSR matrices over z = [W | u | X].  `F` must be 32 bytes (asserted).
pub fn shape<F>(path: &Path, field_id: u32, encoding: u32, num_cons: usize, num_vars: usize, num_io: usize,
                mats: [(&[usize], &[usize], &[F]); 3]) -> Result<()> {
    assert_eq!(size_of::<F>(), 32);
    assert_eq!(size_of::<usize>(), 8);
    let mut w = BufWriter::new(File::create(path)?);
    header(&mut w, KIND_SHAPE, field_id, encoding, num_cons as u64, num_vars as u64, num_io as u64)?;
    for (indptr, indices, data) in mats {
        assert!(indptr.len() == num_cons + 1 && indices.len() == data.len() && indptr[num_cons] == indices.len());
        w.write_all(&(indices.len() as u64).to_le_bytes())?;
        w.write_all(bytes_of(indptr))?;
        w.write_all(bytes_of(indices))?;
        w.write_all(bytes_of(data))?;
    }
    w.flush()
}

/// `CommitmentKey::ck`: affine points of 64 bytes, identity = (0, 0).
pub fn key<A>(path: &Path, curve_id: u32, encoding: u32, points: &[A]) -> Result<()> {
    assert_eq!(size_of::<A>(), 64);
    let mut w = BufWriter::new(File::create(path)?);
    header(&mut w, KIND_KEY, curve_id, encoding, points.len() as u64, 0, 0)?;
    w.write_all(bytes_of(points))?;
    w.flush()
}

/// The fresh (W, X) of consecutive folding steps; the step count in the header is patched when the writer is finished.
pub struct WitnessWriter {
    w: BufWriter<File>,
    num_vars: usize,
    num_io: usize,
    steps: u64,
    head: [u32; 2],
}
impl WitnessWriter {
    /// `pp_digest`: the canonical 32 bytes of `PublicParams::digest()` (`to_repr()`), whatever the encoding of the vectors.
    pub fn create(path: &Path, field_id: u32, encoding: u32, num_vars: usize, num_io: usize, pp_digest: &[u8; 32]) -> Result<Self> {
        let mut w = BufWriter::new(File::create(path)?);
        header(&mut w, KIND_WITNESS, field_id, encoding, num_vars as u64, num_io as u64, 0)?;
        w.write_all(pp_digest)?;
        Ok(Self { w, num_vars, num_io, steps: 0, head: [field_id, encoding] })
    }
    pub fn step<F>(&mut self, w_vec: &[F], x: &[F]) -> Result<()> {
        assert_eq!(size_of::<F>(), 32);
        assert!(w_vec.len() == self.num_vars && x.len() == self.num_io);
        self.w.write_all(bytes_of(w_vec))?;
        self.w.write_all(bytes_of(x))?;
        self.steps += 1;
        Ok(())
    }
    pub fn finish(mut self) -> Result<()> {
        self.w.flush()?;
        let mut f = self.w.into_inner().map_err(|e| e.into_error())?;
        f.seek(SeekFrom::Start(0))?;
        header(&mut f, KIND_WITNESS, self.head[0], self.head[1], self.num_vars as u64, self.num_io as u64, self.steps)?;
        f.flush()
    }
}

/// Kind 4, the probe record: ONE `(transcript inputs -> r)` pair of a real `NIFS::prove` and the first points of the real commitment
/// key.  `python -m lurk_beta_amd.dump probe FILE [--search]` runs the library's transcript and `from_label` over it, names the stage
/// that disagrees and - with `--search` - the `lurk_hip_ro_params` / `lurk_hip_ck_params` fields that reproduce it.  Inside arecibo's
/// `NIFS::prove` (where the values are visible), right after `let r = ro.squeeze(NUM_CHALLENGE_BITS);`:
///
/// ```ignore
/// dump::probe(path, CURVE_PALLAS, ENC_MONTGOMERY, &pp_digest.to_repr(),
///             &U1.comm_W.to_affine(), &U1.comm_E.to_affine(), &U1.u, &U1.X, &U2.comm_W.to_affine(), &U2.X, &comm_T.to_affine(),
///             &scalar_r,                 // r as an element of the SCALAR field (what `RelaxedR1CSInstance::fold` multiplies by)
///             &ro_state,                 // Vec<Base>: the elements the RO held when it squeezed (clone `ro.state` before `squeeze`); may be empty
///             b"ck", &pp.ck_primary.ck[..8])?;
/// ```
/// `A` is a 64-byte affine point `{x, y}` (identity `(0, 0)`), `F` / `B` 32-byte elements of the scalar / base field.
#[allow(clippy::too_many_arguments)]
pub fn probe<A, F, B>(path: &Path, curve_id: u32, encoding: u32, pp_digest: &[u8; 32], comm_w1: &A, comm_e1: &A, u1: &F, x1: &[F], comm_w2: &A,
                      x2: &[F], comm_t: &A, r: &F, absorbed: &[B], label: &[u8], key_points: &[A]) -> Result<()> {
    assert_eq!(size_of::<A>(), 64);
    assert_eq!(size_of::<F>(), 32);
    assert_eq!(size_of::<B>(), 32);
    assert_eq!(x1.len(), x2.len());
    let mut w = BufWriter::new(File::create(path)?);
    header(&mut w, KIND_PROBE, curve_id, encoding, x1.len() as u64, absorbed.len() as u64, key_points.len() as u64)?;
    w.write_all(pp_digest)?;
    w.write_all(bytes_of(std::slice::from_ref(comm_w1)))?;
    w.write_all(bytes_of(std::slice::from_ref(comm_e1)))?;
    w.write_all(bytes_of(std::slice::from_ref(u1)))?;
    w.write_all(bytes_of(x1))?;
    w.write_all(bytes_of(std::slice::from_ref(comm_w2)))?;
    w.write_all(bytes_of(x2))?;
    w.write_all(bytes_of(std::slice::from_ref(comm_t)))?;
    w.write_all(bytes_of(std::slice::from_ref(r)))?;
    w.write_all(bytes_of(absorbed))?;
    w.write_all(&(label.len() as u64).to_le_bytes())?;
    w.write_all(label)?;
    w.write_all(&[0u8; 8][..(8 - label.len() % 8) % 8])?;
    w.write_all(bytes_of(key_points))?;
    w.flush()
}

//! `nova/src/prover.rs:53-90` `compute_cross_term` on the device: T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1 in one fused
//! kernel (`kg_nova_cross_term`), the six sparse products of the reference (`SparseMatrix::prod`, zkstd/src/matrix.rs:32-47)
//! included.  The shape matrices cross the ABI as CSR over z = (u | x | w) -- `SparseMatrix::to_csr`, added by
//! `patches/zkstd_matrix_csr.diff`, applies `prod`'s own index rule (Instance(i) -> i, Witness(i) -> i + l) -- and stay
//! resident: `R1csShape::matrices()` clones them on every call, so the cache is keyed by content (a 128-bit hash over every
//! row boundary, column and coefficient: `csr::content_hash`), not by address; a call whose hash matches the resident copy
//! builds no host CSR and uploads nothing but z1, z2.
use std::collections::HashMap;
use std::sync::Mutex;

use kogarashi_amd_sys as sys;
use zkstd::common::PrimeField;
use zkstd::matrix::SparseMatrix;

use crate::csr::{content_hash, ResidentShape};
use crate::{contexts, scalar_words, DeviceBuf};

// const `Mutex::new` (available on the pinned nightly-2022-11-14; `OnceLock` is not): the map is made on first use
static SHAPES: Mutex<Option<HashMap<(usize, usize, i32), ResidentShape>>> = Mutex::new(None);

/// The cross term of a folding step; `None` (no device, a field the backend does not serve, any non-zero status) lets the
/// caller's CPU body run.  `l` = x.len() + 1 for both pairs (the relaxed instance and the fresh one share the shape).
#[allow(clippy::too_many_arguments)]
pub fn cross_term<F: PrimeField + 'static>(a: &SparseMatrix<F>, b: &SparseMatrix<F>, c: &SparseMatrix<F>, m: u64, l: usize, z1: &[F],
                                           z2: &[F], u1: &F, u2: &F) -> Option<Vec<F>> {
    let (_, field) = scalar_words(z1)?;
    if z1.len() != z2.len() || m == 0 {
        return None;
    }
    let ctxs = contexts()?;
    let ctx = ctxs.lock(0)?;                                   // the shapes are resident on GPU 0
    let ctx = &*ctx;
    // `R1csShape::matrices()` hands every call fresh clones (nova/src/prover.rs:63), so neither an address nor a cached
    // `Parameters`-like entry identifies the shape here: one hashing pass per folding step (128 bits over every row boundary,
    // column and coefficient plus the exact entry counts) -- O(nnz) host work, the same order as the clone the reference's own
    // call site has just made -- decides whether the resident copy is still the shape's
    let print = content_hash([a, b, c], l)?;
    let mut guard = SHAPES.lock().ok()?;
    let shapes = guard.get_or_insert_with(HashMap::new);
    let key = (m as usize, l, field);
    if shapes.get(&key).map(|s| s.fingerprint != print).unwrap_or(true) {
        shapes.insert(key, ResidentShape::build(ctx, a, b, c, l, print)?);
    }
    let shape = shapes.get(&key)?;
    if !shape.covers(m as usize, z1.len()) {
        return None;                                           // row count or a column index outside (m, z): let the CPU body decide
    }
    let words = |s: &[F]| unsafe { core::slice::from_raw_parts(s.as_ptr() as *const u64, 4 * s.len()) };
    let (d1, d2) = (DeviceBuf::from_words(ctx, words(z1)).ok()?, DeviceBuf::from_words(ctx, words(z2)).ok()?);
    let out = DeviceBuf::new(ctx, m as usize * 32).ok()?;
    let (ca, cb, cc) = (shape.m[0].csr(), shape.m[1].csr(), shape.m[2].csr());
    let rc = unsafe {
        sys::kg_nova_cross_term(ctx.raw(), field, &ca, &cb, &cc, m as usize, d1.as_u64(), d2.as_u64(), u1 as *const F as *const u64,
                                u2 as *const F as *const u64, out.as_u64())
    };
    if rc != sys::KG_OK {
        return None;
    }
    let mut t = vec![F::zero(); m as usize];
    out.read_words(unsafe { core::slice::from_raw_parts_mut(t.as_mut_ptr() as *mut u64, 4 * t.len()) }).ok()?;
    Some(t)
}

/// shortest vector whose fold goes to the device: below this the two PCIe trips cost more than the host's multiply-adds
const AXPY_MIN_LEN: usize = 1 << 14;

/// out[i] = a[i] + s * b[i] over min(len) elements: the vector folds of NIFS -- `RelaxedR1csWitness::fold`
/// (nova/src/relaxed_r1cs/witness.rs:56-70: W = W1 + r W2, E = E1 + r T) and `RelaxedR1csInstance::fold`'s x (instance.rs:81-101)
/// -- as one `kg_field_vec_axpy` on the scalar field of either curve of the cycle.
///
/// The vectors cross the bus -- two up, one down: 96 B per element at ~50 GB/s is ~2 ns per element against ~25 ns for the host's
/// Montgomery product and addition, so the transfer-inclusive device path takes about a tenth of the host's time from 2^14
/// elements up (arithmetic from the measured bus rate, not a measurement of the Rust path: no toolchain in the build image).
/// They cannot STAY on the device between folding steps without changing the reference's own types: `RelaxedR1csWitness { w, e }`
/// holds host `DenseVectors` that the IVC driver clones, encodes and compares (`#[derive(Clone, Encode, Decode, PartialEq)]`), so the
/// folded W and E are returned to the host here, and the next step's cross term uploads them again as part of z1 (INTEGRATION.md,
/// "Nova folds").  `None` (no device, short vectors, an unserved field, any status) lets the CPU body run.
pub fn axpy<F: PrimeField + 'static>(a: &[F], s: &F, b: &[F]) -> Option<Vec<F>> {
    let n = a.len().min(b.len());
    if n < AXPY_MIN_LEN || a.len() != b.len() {
        return None;                                           // DenseVectors' own `+` asserts equal lengths: let it
    }
    let (_, field) = scalar_words(a)?;
    let ctxs = contexts()?;
    let ctx = ctxs.lock_any()?;                                // nothing resident is involved: any free GPU
    let ctx = &*ctx;
    let words = |v: &[F]| unsafe { core::slice::from_raw_parts(v.as_ptr() as *const u64, 4 * v.len()) };
    let (da, db) = (DeviceBuf::from_words(ctx, words(a)).ok()?, DeviceBuf::from_words(ctx, words(b)).ok()?);
    // out aliases a (`kg_field_vec_axpy`: "out may alias an input"): no third buffer
    let rc = unsafe { sys::kg_field_vec_axpy(ctx.raw(), field, da.as_u64(), s as *const F as *const u64, db.as_u64(), da.as_u64(), n) };
    if rc != sys::KG_OK {
        return None;
    }
    let mut out = vec![F::zero(); n];
    da.read_words(unsafe { core::slice::from_raw_parts_mut(out.as_mut_ptr() as *mut u64, 4 * n) }).ok()?;
    Some(out)
}

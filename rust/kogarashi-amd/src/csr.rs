//! Sparse matrices of the reference (`zkstd::matrix::SparseMatrix`) as resident CSR on the device: what `kg_nova_cross_term`,
//! `kg_r1cs_prod` and `kg_groth16_prove_r1cs_bn254` take.  `SparseMatrix::to_csr` (patches/zkstd_matrix_csr.diff) applies
//! `prod`'s / `evaluate_with_z`'s own index rule (Instance(i) -> column i, Witness(i) -> column i + l).
use kogarashi_amd_sys as sys;
use zkstd::common::PrimeField;
use zkstd::matrix::SparseMatrix;

use crate::{scalar_words, Context, DeviceBuf};

pub(crate) struct HostCsr {
    pub row_ptr: Vec<u64>,
    pub col: Vec<u64>,
    pub val: Vec<u64>,
}
impl HostCsr {
    /// `SparseMatrix::to_csr` plus the coefficients as the ABI's words
    pub fn of<F: PrimeField + 'static>(mat: &SparseMatrix<F>, l: usize) -> Option<Self> {
        let (row_ptr, col, val) = mat.to_csr(l);
        let (words, _) = scalar_words(&val)?;
        let val = unsafe { core::slice::from_raw_parts(words, 4 * val.len()) }.to_vec();
        Some(Self { row_ptr, col, val })
    }
    /// largest column index (what z must cover), 0 for an empty matrix
    pub fn max_col(&self) -> u64 {
        self.col.iter().copied().max().unwrap_or(0)
    }
    pub fn upload(&self, ctx: &Context) -> Option<ResidentMatrix> {
        let pad = [0u64; 4];
        Some(ResidentMatrix {
            row_ptr: DeviceBuf::from_words(ctx, &self.row_ptr).ok()?,
            col: DeviceBuf::from_words(ctx, if self.col.is_empty() { &pad[..1] } else { &self.col }).ok()?,
            val: DeviceBuf::from_words(ctx, if self.val.is_empty() { &pad[..] } else { &self.val }).ok()?,
        })
    }
}

pub(crate) struct ResidentMatrix {
    row_ptr: DeviceBuf,
    col: DeviceBuf,
    val: DeviceBuf,
}
impl ResidentMatrix {
    pub fn csr(&self) -> sys::KgCsr {
        sys::KgCsr { d_row_ptr: self.row_ptr.as_u64() as *const u64, d_col: self.col.as_u64() as *const u64, d_val: self.val.as_u64() as *const u64 }
    }
}

/// 128-bit content hash of (a, b, c) -- every row boundary, column index and coefficient word goes in -- plus the exact
/// invariants that are cheap to keep (entries per matrix): two constraint systems share resident matrices only if they are
/// the same matrices.  One pass over the entries (`SparseMatrix::for_each_entry`, patches/zkstd_matrix_csr.diff), no
/// allocation.  NOT a cryptographic hash (FNV-1a and a second multiplicative mixer): it guards against a stale cache, not
/// against an adversary who builds colliding circuits.  Structure words (matrix number, row boundary, row count) go through a
/// different mixing step than data words (column, coefficient), so a data word can never stand in for a boundary.
/// Callers pay this pass on FIRST SIGHT of a shape only (groth16.rs binds the shape to the cached `Parameters` entry; nova.rs
/// to the folding shape) -- or on every call under `debug_assertions` / `KOGARASHI_AMD_VERIFY_SHAPE`.
pub(crate) fn content_hash<F: PrimeField + 'static>(mats: [&SparseMatrix<F>; 3], l: usize) -> Option<Fingerprint> {
    scalar_words::<F>(&[])?;                                   // a field the backend serves: 4 x u64 per element
    let (mut h1, mut h2) = (0xcbf2_9ce4_8422_2325u64, 0x9e37_79b9_7f4a_7c15u64);
    let mut nnz = [0u64; 3];
    for (k, mat) in mats.iter().enumerate() {
        let data = |w: u64, h1: &mut u64, h2: &mut u64| {
            *h1 = (*h1 ^ w).wrapping_mul(0x0000_0100_0000_01b3);                                 // FNV-1a over 64-bit words
            *h2 = (h2.rotate_left(23) ^ w).wrapping_mul(0xff51_afd7_ed55_8ccd).wrapping_add(0x2545_f491_4f6c_dd1d);
        };
        let structure = |w: u64, h1: &mut u64, h2: &mut u64| {
            *h1 = (h1.rotate_left(29) ^ w ^ 0xa5a5_a5a5_a5a5_a5a5).wrapping_mul(0x9fb2_1c65_1e98_df25);
            *h2 = (*h2 ^ w.rotate_left(32)).wrapping_mul(0xc2b2_ae3d_27d4_eb4f).wrapping_add(0x1656_67b1_9e37_79f9);
        };
        structure(k as u64, &mut h1, &mut h2);
        let mut last_row = usize::MAX;
        let mut count = 0u64;
        mat.for_each_entry(l, |row, col, coeff| {
            if row != last_row {
                structure(row as u64, &mut h1, &mut h2);       // row boundary (empty rows shift every later row index)
                last_row = row;
            }
            data(col, &mut h1, &mut h2);
            let w = unsafe { &*(coeff as *const F as *const [u64; 4]) };
            w.iter().for_each(|&x| data(x, &mut h1, &mut h2));
            count += 1;
        });
        structure(mat.rows() as u64, &mut h1, &mut h2);
        nnz[k] = count;
    }
    Some(Fingerprint { hash: [h1, h2], nnz, l: l as u64 })
}

/// what identifies the resident copy of a shape: the content hash and the exact counts it was taken over
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub(crate) struct Fingerprint {
    pub hash: [u64; 2],
    pub nnz: [u64; 3],
    pub l: u64,
}

/// Three matrices of one constraint system, resident; `fingerprint` = `content_hash` of the three.
pub(crate) struct ResidentShape {
    pub m: [ResidentMatrix; 3],
    pub fingerprint: Fingerprint,
    /// rows of each matrix and the largest column index any of them holds: what a call's (m, z) must cover
    pub rows: usize,
    pub max_col: u64,
}
unsafe impl Send for ResidentShape {}

impl ResidentShape {
    /// Builds the host CSR of (a, b, c) and uploads it: only when no resident copy carries this fingerprint.
    pub fn build<F: PrimeField + 'static>(ctx: &Context, a: &SparseMatrix<F>, b: &SparseMatrix<F>, c: &SparseMatrix<F>, l: usize,
                                           fingerprint: Fingerprint) -> Option<Self> {
        let host = [HostCsr::of(a, l)?, HostCsr::of(b, l)?, HostCsr::of(c, l)?];
        if (0..3).any(|k| host[k].col.len() as u64 != fingerprint.nnz[k]) {
            return None;                                       // the fingerprint was not taken over these matrices
        }
        let rows = host[0].row_ptr.len() - 1;
        if host.iter().any(|h| h.row_ptr.len() != rows + 1) {
            return None;                                       // the three matrices of a shape have one row per constraint
        }
        let max_col = host.iter().map(|h| h.max_col()).max().unwrap_or(0);
        Some(Self { m: [host[0].upload(ctx)?, host[1].upload(ctx)?, host[2].upload(ctx)?], fingerprint, rows, max_col })
    }
    /// the kernels index row_ptr[0..=m] and z[col]: refuse a call the resident matrices do not cover (the reference would panic)
    pub fn covers(&self, m: usize, z_len: usize) -> bool {
        self.rows == m && (self.max_col as usize) < z_len.max(1)
    }
}

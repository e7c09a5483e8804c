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
    /// sizes plus the first and last entries: the matrices reach the glue as fresh clones, so residency is keyed by content
    pub fn fingerprint(&self, out: &mut Vec<u64>) {
        out.extend([self.row_ptr.len() as u64, self.col.len() as u64]);
        out.extend(self.col.iter().take(8));
        out.extend(self.col.iter().rev().take(8));
        out.extend(self.val.iter().take(16));
        out.extend(self.val.iter().rev().take(16));
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

/// Three matrices of one constraint system, resident; `fingerprint` as `HostCsr::fingerprint` of the three in order.
pub(crate) struct ResidentShape {
    pub m: [ResidentMatrix; 3],
    pub fingerprint: Vec<u64>,
}
unsafe impl Send for ResidentShape {}

impl ResidentShape {
    /// Host CSR of (a, b, c) with its fingerprint; `upload` only when the resident copy (if any) does not match.
    pub fn host<F: PrimeField + 'static>(a: &SparseMatrix<F>, b: &SparseMatrix<F>, c: &SparseMatrix<F>, l: usize) -> Option<([HostCsr; 3], Vec<u64>)> {
        let host = [HostCsr::of(a, l)?, HostCsr::of(b, l)?, HostCsr::of(c, l)?];
        let mut print = Vec::new();
        host.iter().for_each(|h| h.fingerprint(&mut print));
        Some((host, print))
    }
    pub fn upload(ctx: &Context, host: &[HostCsr; 3], fingerprint: Vec<u64>) -> Option<Self> {
        Some(Self { m: [host[0].upload(ctx)?, host[1].upload(ctx)?, host[2].upload(ctx)?], fingerprint })
    }
}

//! Raw bindings of the C ABI in `include/kogarashi_amd.h` (the drop-in boundary of the MI355X backend).
//!
//! `ffi.rs` is generated from the header (`tools/gen_rust_ffi.py`); this file holds the types the declarations use.
//! Field elements cross the boundary as the reference's own in-memory form: `[u64; 4]` little-endian Montgomery limbs
//! (`bn254/src/fr.rs:71`, `bn254/src/fq.rs:48`); points as `x | y` words plus a separate flag byte, because the
//! reference's affine structs are `repr(Rust)` (`bn254/src/g1.rs:18-22`).
#![no_std]
#![allow(non_camel_case_types)]

mod ffi;
pub use ffi::*;

/// `kg_ctx`: one per (process, GPU); calls on one context are serialised by the caller.
#[repr(C)]
pub struct KgCtx {
    _private: [u8; 0],
}

/// `kg_sharded_key`: a commitment key resident across several contexts (`kg_sharded_key_*`).
#[repr(C)]
pub struct KgShardedKey {
    _private: [u8; 0],
}

/// `kg_groth16_crs`, field for field (`groth16/src/params.rs:6-28` resident on the device plus the host-side vk points).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct KgGroth16Crs {
    pub m: usize,
    pub l: usize,
    pub m_l_1: usize,
    pub d_h: *const u64,
    pub d_h_inf: *const u8,
    pub d_l: *const u64,
    pub d_l_inf: *const u8,
    pub d_a: *const u64,
    pub d_a_inf: *const u8,
    pub d_b_g1: *const u64,
    pub d_b_g1_inf: *const u8,
    pub d_b_g2: *const u64,
    pub d_b_g2_inf: *const u8,
    pub alpha_g1: [u64; 8],
    pub beta_g1: [u64; 8],
    pub delta_g1: [u64; 8],
    pub beta_g2: [u64; 16],
    pub delta_g2: [u64; 16],
    pub delta_g1_inf: u8,
    pub delta_g2_inf: u8,
}

/// `kg_csr`: one sparse matrix of an R1CS shape as CSR over z = (u | x | w), device pointers (`kg_nova_cross_term`).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct KgCsr {
    pub d_row_ptr: *const u64,
    pub d_col: *const u64,
    pub d_val: *const u64,
}

// kg_status
pub const KG_OK: i32 = 0;
pub const KG_ERR_NO_DEVICE: i32 = -1;
pub const KG_ERR_BAD_ARG: i32 = -2;
pub const KG_ERR_OOM: i32 = -3;
pub const KG_ERR_HIP: i32 = -4;
pub const KG_ERR_UNSUPPORTED: i32 = -5;
/// delta is the identity: `Error::ProverSubVersionCrsAttack` (`groth16/src/prover.rs:67-69`)
pub const KG_ERR_CRS: i32 = -6;
/// a toxic scalar has no inverse: `Error::ProverInversionFailed` (`groth16/src/zksnark.rs:37-38`)
pub const KG_ERR_INVERSION: i32 = -7;

// field / curve selectors
pub const KG_FR: i32 = 0;
pub const KG_FQ: i32 = 1;
pub const KG_G1: i32 = 0;
pub const KG_GRUMPKIN: i32 = 1;
pub const KG_G2: i32 = 2;

// kg_field_op
pub const KG_OP_ADD: i32 = 0;
pub const KG_OP_SUB: i32 = 1;
pub const KG_OP_MUL: i32 = 2;
pub const KG_OP_SQUARE: i32 = 3;
pub const KG_OP_NEG: i32 = 4;
pub const KG_OP_DOUBLE: i32 = 5;
pub const KG_OP_INVERT: i32 = 6;
pub const KG_OP_FROM_MONT: i32 = 7;
pub const KG_OP_TO_MONT: i32 = 8;

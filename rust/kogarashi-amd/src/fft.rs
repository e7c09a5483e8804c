//! `groth16/src/fft.rs:92-154` on the device: `Fft::<Fr>::{dft, idft, coset_dft, coset_idft, divide_by_z_on_coset}`.
//! Natural order in and out; the caller's vector is zero-padded to n = 2^k exactly like `prepare_fft` (fft.rs:157-162).
//! Twiddles are cached per (k, direction) inside the context -- the reference rebuilds them in `Fft::new` per proof.
use core::ffi::c_void;

use bn_254::Fr;
use kogarashi_amd_sys as sys;
use zkstd::common::FftField;

use crate::{cast_slice, contexts, DeviceBuf};

#[derive(Clone, Copy, PartialEq, Eq)]
pub enum Transform {
    Dft,
    Idft,
    CosetDft,
    CosetIdft,
    DivideByZOnCoset,
}

/// In place on `data` (resized to 2^k).  `None`: not served (F is not bn254's Fr, no device, or a backend error) --
/// `data` is then untouched apart from the resize the CPU body performs anyway.
pub fn transform<F: FftField + 'static>(k: usize, data: &mut Vec<F>, what: Transform) -> Option<()> {
    let n = 1usize << k;
    data.resize(n, F::zero());
    cast_slice::<F, Fr>(data)?;
    let ctxs = contexts()?;
    let ctx = ctxs.lock_any()?;                                // a transform brings its data along: any GPU that is free
    let ctx = &*ctx;
    let words = unsafe { core::slice::from_raw_parts(data.as_ptr() as *const u64, 4 * n) };
    let d = DeviceBuf::from_words(ctx, words).ok()?;
    let rc = unsafe {
        match what {
            Transform::Dft => sys::kg_ntt_bn254_fr(ctx.raw(), d.as_u64(), k as u32, 0, 0),
            Transform::Idft => sys::kg_ntt_bn254_fr(ctx.raw(), d.as_u64(), k as u32, 1, 0),
            Transform::CosetDft => sys::kg_ntt_bn254_fr(ctx.raw(), d.as_u64(), k as u32, 0, 1),
            Transform::CosetIdft => sys::kg_ntt_bn254_fr(ctx.raw(), d.as_u64(), k as u32, 1, 1),
            Transform::DivideByZOnCoset => sys::kg_fr_divide_by_z_on_coset(ctx.raw(), d.as_u64(), k as u32),
        }
    };
    if rc != sys::KG_OK {
        return None;
    }
    let rc = unsafe { sys::kg_memcpy_d2h(ctx.raw(), data.as_mut_ptr() as *mut c_void, d.as_u64() as *const c_void, 32 * n) };
    if rc != sys::KG_OK {
        return None;
    }
    Some(())
}

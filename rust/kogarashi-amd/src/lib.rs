//! Glue between Kogarashi's own types (zkstd traits, bn-254, grumpkin) and the MI355X backend's C ABI.
//!
//! The reference reaches its hot path through crate-internal generic functions:
//!   `groth16/src/msm.rs:6`          `msm_curve_addition<C: BNAffine>(bases, coeffs) -> C::Extended`
//!   `groth16/src/fft.rs:92-154`     `Fft::<F>::{dft, idft, coset_dft, coset_idft, divide_by_z_on_coset}`
//!   `nova/src/pedersen.rs:15`       `PedersenCommitment::<C>::commit(&self, m) -> C`
//!   `groth16/src/prover.rs:20`      `Prover::create_proof`
//!   `nova/src/prover.rs:53`         `Prover::compute_cross_term`
//! The patches in `rust/patches/` add one `#[cfg(feature = "gpu")]` early return to each of them that calls the
//! generic entry points below.  They dispatch on `TypeId` (the `'static` bound comes from the one-line patch to
//! `zkstd::traits::Group`), so the reference's generic signatures stay as they are; a type the backend does not
//! serve (or any non-zero status) yields `None` and the crate's own CPU body runs -- the *library* has no CPU path.
//!
//! Marshalling: field elements are passed as they lie in memory (`Fr(pub [u64; 4])`, `Fq`, made
//! `#[repr(transparent)]` by `patches/bn254.diff`); points are `repr(Rust)` structs and are marshalled explicitly
//! through `get_x() / get_y() / is_identity()` into `x | y` words plus a flag byte.
use core::any::TypeId;
use core::ffi::c_void;
use core::ptr;

use bn_254::{Fq, Fq2, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use grumpkin::{Affine as GkAffine, Projective as GkProjective};
use kogarashi_amd_sys as sys;
use zkstd::common::{BNAffine, BNProjective, CurveGroup, Group};

mod csr;
pub mod fft;
pub mod groth16;
pub mod nova;
pub mod pedersen;

/// A backend failure; callers in the patched crates treat it as "use the CPU body".
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct Status(pub i32);

fn check(rc: i32) -> Result<(), Status> {
    if rc == sys::KG_OK {
        Ok(())
    } else {
        Err(Status(rc))
    }
}

/// One `kg_ctx` (a GPU, its streams, twiddle caches and MSM work space).  `Send`: the reference's types are
/// `Send + Sync` and the ABI allows any thread; calls on one context are serialised by `&mut self` / the mutex below.
pub struct Context {
    raw: *mut sys::KgCtx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Self, Status> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::kg_ctx_create(device, &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn raw(&self) -> *mut sys::KgCtx {
        self.raw
    }
    pub fn device_count() -> i32 {
        unsafe { sys::kg_device_count() }
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::kg_ctx_destroy(self.raw) }
    }
}

/// Device memory owned through the ABI's own allocator (a Rust host never links the HIP runtime).
pub struct DeviceBuf {
    ctx: *mut sys::KgCtx,
    ptr: *mut c_void,
    bytes: usize,
}
impl DeviceBuf {
    pub fn new(ctx: &Context, bytes: usize) -> Result<Self, Status> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::kg_malloc(ctx.raw, bytes.max(1), &mut p) })?;
        Ok(Self { ctx: ctx.raw, ptr: p, bytes })
    }
    pub fn from_words(ctx: &Context, w: &[u64]) -> Result<Self, Status> {
        let b = Self::new(ctx, w.len() * 8)?;
        check(unsafe { sys::kg_memcpy_h2d(ctx.raw, b.ptr, w.as_ptr() as *const c_void, w.len() * 8) })?;
        Ok(b)
    }
    pub fn from_bytes(ctx: &Context, w: &[u8]) -> Result<Self, Status> {
        let b = Self::new(ctx, w.len())?;
        check(unsafe { sys::kg_memcpy_h2d(ctx.raw, b.ptr, w.as_ptr() as *const c_void, w.len()) })?;
        Ok(b)
    }
    pub fn read_words(&self, out: &mut [u64]) -> Result<(), Status> {
        debug_assert!(out.len() * 8 <= self.bytes);
        check(unsafe { sys::kg_memcpy_d2h(self.ctx, out.as_mut_ptr() as *mut c_void, self.ptr, out.len() * 8) })
    }
    pub fn as_u64(&self) -> *mut u64 {
        self.ptr as *mut u64
    }
    pub fn as_u8(&self) -> *mut u8 {
        self.ptr as *mut u8
    }
}
impl Drop for DeviceBuf {
    fn drop(&mut self) {
        unsafe { sys::kg_free(self.ctx, self.ptr) };
    }
}

mod global {
    use super::*;
    use std::sync::{Mutex, MutexGuard};

    // `Mutex::new` is const on the reference's pinned toolchain (nightly-2022-11-14); `OnceLock` is not stable there.
    // None: not probed yet; Some(vec): probed -- an empty vec means "no device", and is never probed again.
    static CTXS: Mutex<Option<Vec<Context>>> = Mutex::new(None);

    /// The contexts of the visible GPUs (one each), created on first use, behind the process-wide lock.
    pub struct Contexts(MutexGuard<'static, Option<Vec<Context>>>);
    impl core::ops::Deref for Contexts {
        type Target = Vec<Context>;
        fn deref(&self) -> &Vec<Context> {
            self.0.as_ref().unwrap()
        }
    }

    /// One context per visible GPU, created on first use; `None` when no device is present (CPU bodies run).
    pub fn contexts() -> Option<Contexts> {
        let mut guard = CTXS.lock().ok()?;
        if guard.is_none() {
            let mut v = Vec::new();
            // one hardware queue per library queue; a no-op when the host exported GPU_MAX_HW_QUEUES or initialised HIP itself.
            // The glue's first backend call is the host's opt-in (`gpu` feature): call `kogarashi_amd::contexts()` from the
            // start-up path if other threads may be reading the environment later.
            unsafe { sys::kg_init() };
            let n = Context::device_count();
            for d in 0..n.max(0) {
                match Context::new(d) {
                    Ok(c) => v.push(c),
                    Err(_) => {
                        v.clear();
                        break;
                    }
                }
            }
            *guard = Some(v);
        }
        if guard.as_ref().map(|v| v.is_empty()).unwrap_or(true) {
            return None;
        }
        Some(Contexts(guard))
    }
}
pub use global::contexts;

// ---- TypeId casts (same type on both sides, so these are identity conversions) ------------------------------------
pub(crate) fn same<A: 'static, B: 'static>() -> bool {
    TypeId::of::<A>() == TypeId::of::<B>()
}
pub(crate) fn cast_slice<A: 'static, B: 'static>(s: &[A]) -> Option<&[B]> {
    if same::<A, B>() {
        Some(unsafe { &*(s as *const [A] as *const [B]) })
    } else {
        None
    }
}
pub(crate) fn cast_val<A: 'static + Copy, B: 'static + Copy>(a: A) -> Option<B> {
    if same::<A, B>() {
        Some(unsafe { *(&a as *const A as *const B) })
    } else {
        None
    }
}

/// `&[Fr]` / `&[Fq]` as the ABI's n x 4 words (the structs are `repr(transparent)` over `[u64; 4]`).
pub(crate) fn scalar_words<S: 'static>(s: &[S]) -> Option<(*const u64, i32)> {
    const _: () = assert!(core::mem::size_of::<Fr>() == 32 && core::mem::size_of::<Fq>() == 32);
    if same::<S, Fr>() {
        Some((s.as_ptr() as *const u64, sys::KG_FR))
    } else if same::<S, Fq>() {
        Some((s.as_ptr() as *const u64, sys::KG_FQ))
    } else {
        None
    }
}

/// The three curves of the path, with their ABI marshalling.
pub trait GpuCurve: BNAffine + Copy + 'static {
    const CURVE: i32;
    /// u64 words of one affine point (x | y)
    const WORDS: usize;
    fn put_xy(&self, out: &mut Vec<u64>);
    fn affine_from(words: &[u64], inf: bool) -> Self;
    /// ABI projective output (x, y, z) with z in {0, 1}
    fn extended_from(words: &[u64]) -> Self::Extended;
}

fn fq(w: &[u64]) -> Fq {
    Fq::new_unchecked([w[0], w[1], w[2], w[3]])
}
fn fr(w: &[u64]) -> Fr {
    Fr::new_unchecked([w[0], w[1], w[2], w[3]])
}
fn fq2(w: &[u64]) -> Fq2 {
    Fq2::new_unchecked([fq(&w[0..4]), fq(&w[4..8])])
}

impl GpuCurve for G1Affine {
    const CURVE: i32 = sys::KG_G1;
    const WORDS: usize = 8;
    fn put_xy(&self, out: &mut Vec<u64>) {
        out.extend_from_slice(self.get_x().inner());
        out.extend_from_slice(self.get_y().inner());
    }
    fn affine_from(w: &[u64], inf: bool) -> Self {
        if inf {
            G1Affine::ADDITIVE_IDENTITY
        } else {
            G1Affine::from_x_and_y(fq(&w[0..4]), fq(&w[4..8]))
        }
    }
    fn extended_from(w: &[u64]) -> G1Projective {
        G1Projective::new_unchecked(fq(&w[0..4]), fq(&w[4..8]), fq(&w[8..12]))
    }
}
impl GpuCurve for GkAffine {
    const CURVE: i32 = sys::KG_GRUMPKIN;
    const WORDS: usize = 8;
    fn put_xy(&self, out: &mut Vec<u64>) {
        out.extend_from_slice(self.get_x().inner());
        out.extend_from_slice(self.get_y().inner());
    }
    fn affine_from(w: &[u64], inf: bool) -> Self {
        if inf {
            GkAffine::ADDITIVE_IDENTITY
        } else {
            GkAffine::from_x_and_y(fr(&w[0..4]), fr(&w[4..8]))
        }
    }
    fn extended_from(w: &[u64]) -> GkProjective {
        GkProjective::new_unchecked(fr(&w[0..4]), fr(&w[4..8]), fr(&w[8..12]))
    }
}
impl GpuCurve for G2Affine {
    const CURVE: i32 = sys::KG_G2;
    const WORDS: usize = 16;
    fn put_xy(&self, out: &mut Vec<u64>) {
        let (x, y) = (self.get_x(), self.get_y());
        for c in x.inner().iter().chain(y.inner().iter()) {
            out.extend_from_slice(c.inner()); // Fq2::inner: patches/bn254.diff
        }
    }
    fn affine_from(w: &[u64], inf: bool) -> Self {
        if inf {
            G2Affine::ADDITIVE_IDENTITY
        } else {
            G2Affine::from_x_and_y(fq2(&w[0..8]), fq2(&w[8..16]))
        }
    }
    fn extended_from(w: &[u64]) -> G2Projective {
        G2Projective::new_unchecked(fq2(&w[0..8]), fq2(&w[8..16]), fq2(&w[16..24]))
    }
}

/// x | y words and flag bytes of a point slice
pub(crate) fn marshal<C: GpuCurve>(pts: &[C]) -> (Vec<u64>, Vec<u8>) {
    let mut xy = Vec::with_capacity(pts.len() * C::WORDS);
    let mut inf = Vec::with_capacity(pts.len());
    for p in pts {
        p.put_xy(&mut xy);
        inf.push(p.is_identity() as u8);
    }
    (xy, inf)
}

fn msm_typed<C: GpuCurve>(bases: &[C], coeffs: *const u64, n: usize) -> Option<C::Extended> {
    let ctxs = contexts()?;
    let (xy, inf) = marshal(&bases[..n]);
    let mut out = [0u64; 24];
    let rc = unsafe { sys::kg_msm_host(ctxs[0].raw(), C::CURVE, xy.as_ptr(), inf.as_ptr(), coeffs, n, out.as_mut_ptr()) };
    if rc != sys::KG_OK {
        return None;
    }
    Some(C::extended_from(&out))
}

/// `groth16::msm::msm_curve_addition` on the device: sum over `min(len)` pairs (the reference zips, msm.rs:25).
/// Returns `(x, y, 1)` / `(0, 1, 0)`; equal to the CPU result under the crate's projective `PartialEq`
/// (cross-multiplication, macros/curve/weierstrass/group.rs:89-97) and bit for bit after `.into()`.
pub fn msm<C: BNAffine + 'static>(bases: &[C], coeffs: &[C::Scalar]) -> Option<C::Extended>
where
    C::Scalar: 'static,
    C::Extended: 'static + Copy,
{
    let n = bases.len().min(coeffs.len());
    let (sw, _) = scalar_words(coeffs)?;
    if let Some(b) = cast_slice::<C, G1Affine>(bases) {
        return cast_val(msm_typed(b, sw, n)?);
    }
    if let Some(b) = cast_slice::<C, G2Affine>(bases) {
        return cast_val(msm_typed(b, sw, n)?);
    }
    if let Some(b) = cast_slice::<C, GkAffine>(bases) {
        return cast_val(msm_typed(b, sw, n)?);
    }
    None
}

//! `groth16/src/prover.rs:20-99` on the device: the CRS (`Parameters`, params.rs:6-28) is uploaded and converted to
//! the MSM's internal form once per `Prover` (it is immutable), and so are the constraint matrices of the circuit; a proof
//! uploads `cs.x()`, `cs.w()`, runs `cs.evaluate()` (three sparse products), 7 transforms, the fused h = (a o b - c) / Z,
//! five MSMs (the reference's eight, merged) and the host assembly inside `kg_groth16_prove_r1cs_bn254`, and returns the
//! three affine points.  (`prove` takes host-side evaluation vectors instead.)
use std::collections::HashMap;
use std::sync::{Arc, Mutex};

use bn_254::{Fr, G1Affine, G2Affine};
use kogarashi_amd_sys as sys;
use zkstd::common::CurveGroup;
use zkstd::matrix::SparseMatrix;

use crate::csr::{content_hash, ResidentShape};
use crate::{contexts, marshal, DeviceBuf, GpuCurve, Status};

/// Device-resident `Parameters` (+ the vk points the assembly needs).  Built once, e.g. in `Prover::new`/first proof.
pub struct ResidentCrs {
    crs: sys::KgGroth16Crs,
    _bufs: Vec<DeviceBuf>,
    /// the circuit's constraint matrices, resident as CSR (uploaded by the first `prove_cs`, replaced when the content changes)
    shape: Mutex<Option<ResidentShape>>,
}
unsafe impl Send for ResidentCrs {}
unsafe impl Sync for ResidentCrs {}

fn upload<C: GpuCurve>(ctx: &crate::Context, pts: &[C], bufs: &mut Vec<DeviceBuf>) -> Result<(*const u64, *const u8), Status> {
    let (xy, inf) = marshal(pts);
    let has_inf = inf.iter().any(|&f| f != 0);
    let d = DeviceBuf::from_words(ctx, &xy)?;
    let p = d.as_u64() as *const u64;
    bufs.push(d);
    let mut pi = core::ptr::null();
    if has_inf {
        let di = DeviceBuf::from_bytes(ctx, &inf)?;
        pi = di.as_u8() as *const u8;
        bufs.push(di);
    }
    let rc = unsafe { sys::kg_bases_register(ctx.raw(), C::CURVE, p, pi, pts.len()) };
    if rc != sys::KG_OK {
        return Err(Status(rc));
    }
    Ok((p, pi))
}

fn words8(p: &G1Affine) -> [u64; 8] {
    let mut v = Vec::with_capacity(8);
    p.put_xy(&mut v);
    v.try_into().unwrap()
}
fn words16(p: &G2Affine) -> [u64; 16] {
    let mut v = Vec::with_capacity(16);
    p.put_xy(&mut v);
    v.try_into().unwrap()
}

impl ResidentCrs {
    /// h, l, a, b_g1, b_g2 of `Parameters`; alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2 of its `VerifyingKey`;
    /// (m, l, m_l_1) = (cs.m(), cs.l(), cs.m_l_1()) of the circuit the CRS was made for.
    #[allow(clippy::too_many_arguments)]
    pub fn new(h: &[G1Affine], l: &[G1Affine], a: &[G1Affine], b_g1: &[G1Affine], b_g2: &[G2Affine], alpha_g1: &G1Affine,
               beta_g1: &G1Affine, delta_g1: &G1Affine, beta_g2: &G2Affine, delta_g2: &G2Affine, m: usize, n_inputs: usize,
               n_aux: usize) -> Result<Self, Status> {
        let ctxs = contexts().ok_or(Status(sys::KG_ERR_NO_DEVICE))?;
        let ctx = &ctxs[0];
        let mut bufs = Vec::new();
        let (d_h, d_h_inf) = upload(ctx, h, &mut bufs)?;
        let (d_l, d_l_inf) = upload(ctx, l, &mut bufs)?;
        let (d_a, d_a_inf) = upload(ctx, a, &mut bufs)?;
        let (d_b_g1, d_b_g1_inf) = upload(ctx, b_g1, &mut bufs)?;
        let (d_b_g2, d_b_g2_inf) = upload(ctx, b_g2, &mut bufs)?;
        // Window tables (kg_bases_precompute: 2^(17 w) * P for every window, 15 x the CRS in device memory, built once): each of
        // the proof's five MSMs then uses one bucket set.  Offered for 2^16 .. 2^20 scalars; KOGARASHI_AMD_NO_TABLES=1 keeps the
        // plain resident form.  l meets the whole witness vector z = x || w, h its own m - 1 coefficients.
        let nz = n_inputs + n_aux;
        let in_range = |v: usize| (1usize << 16..=1usize << 20).contains(&v);
        if std::env::var_os("KOGARASHI_AMD_NO_TABLES").is_none() && in_range(nz) && in_range(h.len()) && h.len() + 1 == m {
            for (p, len) in [(d_a, nz), (d_b_g1, nz), (d_b_g2, nz), (d_l, nz), (d_h, h.len())] {
                let rc = unsafe { sys::kg_bases_precompute(ctx.raw(), p, len) };
                if rc != sys::KG_OK {
                    return Err(Status(rc));
                }
            }
        }
        let crs = sys::KgGroth16Crs {
            m, l: n_inputs, m_l_1: n_aux, d_h, d_h_inf, d_l, d_l_inf, d_a, d_a_inf, d_b_g1, d_b_g1_inf, d_b_g2, d_b_g2_inf,
            alpha_g1: words8(alpha_g1), beta_g1: words8(beta_g1), delta_g1: words8(delta_g1),
            beta_g2: words16(beta_g2), delta_g2: words16(delta_g2),
            delta_g1_inf: delta_g1.is_identity() as u8, delta_g2_inf: delta_g2.is_identity() as u8,
        };
        Ok(Self { crs, _bufs: bufs, shape: Mutex::new(None) })
    }

    /// One proof from the constraint system itself: (a, b, c) = cs.matrices(), x = cs.x(), w = cs.w(); `cs.evaluate()` runs on
    /// the device (each transform chain starts with its matrix-vector product).  (r, s) as for `prove`.
    #[allow(clippy::too_many_arguments)]
    pub fn prove_cs(&self, a: &SparseMatrix<Fr>, b: &SparseMatrix<Fr>, c: &SparseMatrix<Fr>, x: &[Fr], w: &[Fr], r: &Fr, s: &Fr)
                    -> Result<(G1Affine, G2Affine, G1Affine), Status> {
        let ctxs = contexts().ok_or(Status(sys::KG_ERR_NO_DEVICE))?;
        let ctx = &ctxs[0];
        let mut shape = self.shape.lock().map_err(|_| Status(sys::KG_ERR_BAD_ARG))?;
        // one hashing pass over the entries; the host CSR is rebuilt (and uploaded) only when the circuit changed
        let print = content_hash([a, b, c], x.len()).ok_or(Status(sys::KG_ERR_BAD_ARG))?;
        if shape.as_ref().map(|s| s.fingerprint != print).unwrap_or(true) {
            *shape = Some(ResidentShape::build(ctx, a, b, c, x.len(), print).ok_or(Status(sys::KG_ERR_OOM))?);
        }
        if !shape.as_ref().unwrap().covers(self.crs.m, x.len() + w.len()) {
            return Err(Status(sys::KG_ERR_BAD_ARG));
        }
        let m3 = &shape.as_ref().unwrap().m;
        let (ca, cb, cc) = (m3[0].csr(), m3[1].csr(), m3[2].csr());
        let up = |v: &[Fr]| DeviceBuf::from_words(ctx, unsafe { core::slice::from_raw_parts(v.as_ptr() as *const u64, 4 * v.len()) });
        let (dx, dw) = (up(x)?, up(w)?);
        let mut out = [0u64; 32];
        let mut inf = [0u8; 3];
        let rc = unsafe {
            sys::kg_groth16_prove_r1cs_bn254(ctx.raw(), &self.crs, &ca, &cb, &cc, dx.as_u64(), dw.as_u64(), r.inner().as_ptr(),
                                             s.inner().as_ptr(), out.as_mut_ptr(), inf.as_mut_ptr())
        };
        if rc != sys::KG_OK {
            return Err(Status(rc));
        }
        Ok((G1Affine::affine_from(&out[0..8], inf[0] != 0), G2Affine::affine_from(&out[8..24], inf[1] != 0),
            G1Affine::affine_from(&out[24..32], inf[2] != 0)))
    }

    /// One proof: (a, b, c) = cs.evaluate(), x = cs.x(), w = cs.w(), (r, s) drawn by the caller from its rng exactly
    /// as prover.rs:71-72 does.  `Err(Status(KG_ERR_CRS))` is `Error::ProverSubVersionCrsAttack` (prover.rs:67-69).
    pub fn prove(&self, a: &[Fr], b: &[Fr], c: &[Fr], x: &[Fr], w: &[Fr], r: &Fr, s: &Fr) -> Result<(G1Affine, G2Affine, G1Affine), Status> {
        let ctxs = contexts().ok_or(Status(sys::KG_ERR_NO_DEVICE))?;
        let ctx = &ctxs[0];
        let up = |v: &[Fr]| DeviceBuf::from_words(ctx, unsafe { core::slice::from_raw_parts(v.as_ptr() as *const u64, 4 * v.len()) });
        let (da, db, dc, dx, dw) = (up(a)?, up(b)?, up(c)?, up(x)?, up(w)?);
        let mut out = [0u64; 32];
        let mut inf = [0u8; 3];
        let rc = unsafe {
            sys::kg_groth16_prove_bn254(ctx.raw(), &self.crs, da.as_u64(), db.as_u64(), dc.as_u64(), dx.as_u64(), dw.as_u64(),
                                        r.inner().as_ptr(), s.inner().as_ptr(), out.as_mut_ptr(), inf.as_mut_ptr())
        };
        if rc != sys::KG_OK {
            return Err(Status(rc));
        }
        Ok((G1Affine::affine_from(&out[0..8], inf[0] != 0), G2Affine::affine_from(&out[8..24], inf[1] != 0),
            G1Affine::affine_from(&out[24..32], inf[2] != 0)))
    }
}

/// The resident CRS of a `Parameters` value, uploaded on first use.  `groth16::Prover { params }` is built by struct
/// literal in zksnark.rs:126, so the device handle is cached here (keyed by the address and length of `params.h`, guarded
/// by its first and last point) instead of in a new field.
#[allow(clippy::too_many_arguments)]
pub fn resident(h: &[G1Affine], l: &[G1Affine], a: &[G1Affine], b_g1: &[G1Affine], b_g2: &[G2Affine], alpha_g1: &G1Affine,
                beta_g1: &G1Affine, delta_g1: &G1Affine, beta_g2: &G2Affine, delta_g2: &G2Affine, m: usize, n_inputs: usize,
                n_aux: usize) -> Option<Arc<ResidentCrs>> {
    // const `Mutex::new` (the pinned nightly-2022-11-14 has no stable `OnceLock`); a failed upload is cached as `None` so that
    // later proofs go straight to the CPU path instead of repeating the CRS upload and the table build
    static CACHE: Mutex<Option<HashMap<(usize, usize), (Option<Arc<ResidentCrs>>, [u64; 16], usize)>>> = Mutex::new(None);
    if a.is_empty() {
        return None;
    }
    let guard = {
        let mut g = [0u64; 16];
        g[..8].copy_from_slice(&words8(&a[0]));
        g[8..].copy_from_slice(&words8(&a[a.len() - 1]));
        g
    };
    let key = (a.as_ptr() as usize, a.len());
    let mut lock = CACHE.lock().ok()?;
    let cache = lock.get_or_insert_with(HashMap::new);
    if let Some((crs, g, cm)) = cache.get(&key) {
        if *g == guard && *cm == m {
            return crs.clone();
        }
    }
    let crs = ResidentCrs::new(h, l, a, b_g1, b_g2, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, m, n_inputs, n_aux).ok().map(Arc::new);
    cache.insert(key, (crs.clone(), guard, m));
    crs
}

impl Drop for ResidentCrs {
    fn drop(&mut self) {
        if let Some(ctxs) = contexts() {
            for p in [self.crs.d_h, self.crs.d_l, self.crs.d_a, self.crs.d_b_g1, self.crs.d_b_g2] {
                unsafe { sys::kg_bases_unregister(ctxs[0].raw(), p) };
            }
        }
    }
}

//! `groth16/src/prover.rs:20-99` on the device: the CRS (`Parameters`, params.rs:6-28) is uploaded and converted to
//! the MSM's internal form once per `Prover` (it is immutable), and so are the constraint matrices of the circuit; a proof
//! uploads `cs.x()`, `cs.w()`, runs `cs.evaluate()` (three sparse products), 7 transforms, the fused h = (a o b - c) / Z,
//! five MSMs (the reference's eight, merged) and the host assembly inside `kg_groth16_prove_r1cs_bn254`, and returns the
//! three affine points.  (`prove` takes host-side evaluation vectors instead.)
//!
//! One GPU per proof is the default: a node's GPUs then serve as many provers (threads) as there are devices, and a proof takes
//! one context's lock.  `KOGARASHI_AMD_SHARD_GPUS=n` (2 or 3) OPTS IN to splitting ONE proof by task over n GPUs
//! (`kg_groth16_prove_sharded`, SURVEY.md 8e -- the MSMs of prover.rs:51-65 are independent until the assembly and the G2 query is
//! the long pole): GPU 0 holds `b_g2` and runs that query, GPU 1 % n holds `a`, `b_g1`, `l` (the three G1 queries against
//! z = x || w), GPU 2 % n holds `h` and runs the transforms and h's MSM; each CRS vector is uploaded only where it is used, and a
//! proof locks exactly those n contexts (`Contexts::lock_set`), never the rest of the node.  The sharded path is unmeasured on
//! multi-GPU hardware -- hence opt-in.
//!
//! Host cost per proof (INTEGRATION.md has the table): the constraint matrices are bound to the cached `Parameters` entry -- a
//! CRS is made for ONE circuit, so a proof against a resident CRS reuses the resident matrices without looking at them; the
//! O(nnz) content hash (and the `cs.matrices()` clone it needs) runs on first sight only, or on every proof under
//! `debug_assertions` / `KOGARASHI_AMD_VERIFY_SHAPE=1`.
use std::collections::HashMap;
use std::sync::{Arc, Mutex};

use bn_254::{Fr, G1Affine, G2Affine};
use kogarashi_amd_sys as sys;
use zkstd::common::CurveGroup;
use zkstd::matrix::SparseMatrix;

use crate::csr::{content_hash, ResidentShape};
use crate::{contexts, marshal, Context, DeviceBuf, GpuCurve, Status};

/// which context runs which part of a proof: [G2 query, G1 queries, transforms + h]
fn owners(n_ctx: usize) -> [usize; 3] {
    [0, 1 % n_ctx, 2 % n_ctx]
}

/// Device-resident `Parameters` (+ the vk points the assembly needs).  Built once, e.g. in `Prover::new`/first proof.
pub struct ResidentCrs {
    /// one struct per context used (its own device pointers; vectors a context does not hold stay null)
    crs: Vec<sys::KgGroth16Crs>,
    /// contexts the proof is spread over (1: everything on GPU 0)
    n_ctx: usize,
    registered: Vec<(usize, *const u64)>,
    _bufs: Vec<DeviceBuf>,
    /// the circuit's constraint matrices, resident as CSR on the context that runs the transforms (uploaded by the first
    /// `prove_cs`; the CRS is circuit-specific, so later proofs reuse them unverified -- see the module comment)
    shape: Mutex<Option<ResidentShape>>,
}
unsafe impl Send for ResidentCrs {}
unsafe impl Sync for ResidentCrs {}

fn upload<C: GpuCurve>(ctx: &Context, pts: &[C], bufs: &mut Vec<DeviceBuf>) -> Result<(*const u64, *const u8), Status> {
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

/// contexts ONE proof is split over: `KOGARASHI_AMD_SHARD_GPUS` = 2 or 3 opts in, anything else (and unset) is 1
fn shard_gpus() -> usize {
    match std::env::var("KOGARASHI_AMD_SHARD_GPUS").ok().and_then(|v| v.parse::<usize>().ok()) {
        Some(n) if (2..=3).contains(&n) => n,
        _ => 1,
    }
}

fn verify_shape_every_proof() -> bool {
    cfg!(debug_assertions) || std::env::var_os("KOGARASHI_AMD_VERIFY_SHAPE").is_some()
}

impl ResidentCrs {
    /// h, l, a, b_g1, b_g2 of `Parameters`; alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2 of its `VerifyingKey`;
    /// (m, l, m_l_1) = (cs.m(), cs.l(), cs.m_l_1()) of the circuit the CRS was made for.
    #[allow(clippy::too_many_arguments)]
    pub fn new(h: &[G1Affine], l: &[G1Affine], a: &[G1Affine], b_g1: &[G1Affine], b_g2: &[G2Affine], alpha_g1: &G1Affine,
               beta_g1: &G1Affine, delta_g1: &G1Affine, beta_g2: &G2Affine, delta_g2: &G2Affine, m: usize, n_inputs: usize,
               n_aux: usize) -> Result<Self, Status> {
        let ctxs = contexts().ok_or(Status(sys::KG_ERR_NO_DEVICE))?;
        // one GPU per proof unless the host opts in to the task split (see the module comment)
        let n_ctx = shard_gpus().min(ctxs.len()).max(1);
        let own = owners(n_ctx);
        let (_, guards) = ctxs.lock_set(&own).ok_or(Status(sys::KG_ERR_BAD_ARG))?;      // contexts 0 .. n_ctx, in index order
        let mut bufs = Vec::new();
        let mut registered = Vec::new();
        let null8 = core::ptr::null::<u64>();
        let nullb = core::ptr::null::<u8>();
        let blank = sys::KgGroth16Crs {
            m, l: n_inputs, m_l_1: n_aux, d_h: null8, d_h_inf: nullb, d_l: null8, d_l_inf: nullb, d_a: null8, d_a_inf: nullb,
            d_b_g1: null8, d_b_g1_inf: nullb, d_b_g2: null8, d_b_g2_inf: nullb,
            alpha_g1: words8(alpha_g1), beta_g1: words8(beta_g1), delta_g1: words8(delta_g1),
            beta_g2: words16(beta_g2), delta_g2: words16(delta_g2),
            delta_g1_inf: delta_g1.is_identity() as u8, delta_g2_inf: delta_g2.is_identity() as u8,
        };
        let mut crs = vec![blank; n_ctx];
        // each vector goes to the context that runs its query: b_g2 | a, b_g1, l | h
        {
            let c = &*guards[own[0]];
            let (p, pi) = upload(c, b_g2, &mut bufs)?;
            crs[own[0]].d_b_g2 = p;
            crs[own[0]].d_b_g2_inf = pi;
            registered.push((own[0], p));
        }
        {
            let c = &*guards[own[1]];
            let (pa, pai) = upload(c, a, &mut bufs)?;
            let (pb, pbi) = upload(c, b_g1, &mut bufs)?;
            let (pl, pli) = upload(c, l, &mut bufs)?;
            let k = &mut crs[own[1]];
            k.d_a = pa; k.d_a_inf = pai; k.d_b_g1 = pb; k.d_b_g1_inf = pbi; k.d_l = pl; k.d_l_inf = pli;
            registered.extend([(own[1], pa), (own[1], pb), (own[1], pl)]);
        }
        {
            let c = &*guards[own[2]];
            let (p, pi) = upload(c, h, &mut bufs)?;
            crs[own[2]].d_h = p;
            crs[own[2]].d_h_inf = pi;
            registered.push((own[2], p));
        }
        // Window tables (kg_bases_precompute: 2^(17 w) * P for every window, 15 x the CRS in device memory, built once): each of
        // the proof's five MSMs then uses one bucket set.  Offered for 2^16 .. 2^20 scalars; KOGARASHI_AMD_NO_TABLES=1 keeps the
        // plain resident form.  l meets the whole witness vector z = x || w, h its own m - 1 coefficients.
        let nz = n_inputs + n_aux;
        let in_range = |v: usize| (1usize << 16..=1usize << 20).contains(&v);
        if std::env::var_os("KOGARASHI_AMD_NO_TABLES").is_none() && in_range(nz) && in_range(h.len()) && h.len() + 1 == m {
            let k1 = crs[own[1]];
            for (ci, p, len) in [(own[1], k1.d_a, nz), (own[1], k1.d_b_g1, nz), (own[0], crs[own[0]].d_b_g2, nz), (own[1], k1.d_l, nz),
                                 (own[2], crs[own[2]].d_h, h.len())] {
                let rc = unsafe { sys::kg_bases_precompute(guards[ci].raw(), p, len) };
                if rc != sys::KG_OK {
                    return Err(Status(rc));
                }
            }
        }
        Ok(Self { crs, n_ctx, registered, _bufs: bufs, shape: Mutex::new(None) })
    }

    /// the resident matrices for this proof: built on first sight (or when verification finds them stale), reused otherwise
    fn shape_for<'a>(&'a self, ctx: &Context, mats: &mut dyn FnMut() -> (SparseMatrix<Fr>, SparseMatrix<Fr>, SparseMatrix<Fr>), l: usize,
                     z_len: usize) -> Result<std::sync::MutexGuard<'a, Option<ResidentShape>>, Status> {
        let mut shape = self.shape.lock().map_err(|_| Status(sys::KG_ERR_BAD_ARG))?;
        if shape.is_none() || verify_shape_every_proof() {
            let (a, b, c) = mats();
            // one hashing pass over the entries; the host CSR is rebuilt (and uploaded) only when the content differs
            let print = content_hash([&a, &b, &c], l).ok_or(Status(sys::KG_ERR_BAD_ARG))?;
            if shape.as_ref().map(|s| s.fingerprint != print).unwrap_or(true) {
                *shape = Some(ResidentShape::build(ctx, &a, &b, &c, l, print).ok_or(Status(sys::KG_ERR_OOM))?);
            }
        }
        if !shape.as_ref().unwrap().covers(self.crs[0].m, z_len) {
            return Err(Status(sys::KG_ERR_BAD_ARG));
        }
        Ok(shape)
    }

    /// One proof from the constraint system itself.  `matrices` is `|| cs.matrices()`: it is CALLED only when the resident
    /// copy is missing or is being verified (the reference's `matrices()` clones all three, O(nnz)); x = cs.x(), w = cs.w();
    /// `cs.evaluate()` runs on the device (each transform chain starts with its matrix-vector product).  (r, s) as for `prove`.
    pub fn prove_cs_with(&self, mut matrices: impl FnMut() -> (SparseMatrix<Fr>, SparseMatrix<Fr>, SparseMatrix<Fr>), x: &[Fr], w: &[Fr],
                         r: &Fr, s: &Fr) -> Result<(G1Affine, G2Affine, G1Affine), Status> {
        let ctxs = contexts().ok_or(Status(sys::KG_ERR_NO_DEVICE))?;
        let words = |v: &[Fr]| unsafe { core::slice::from_raw_parts(v.as_ptr() as *const u64, 4 * v.len()) };
        if self.n_ctx == 1 {
            let ctx = ctxs.lock(0).ok_or(Status(sys::KG_ERR_BAD_ARG))?;
            let shape = self.shape_for(&ctx, &mut matrices, x.len(), x.len() + w.len())?;
            let m3 = &shape.as_ref().unwrap().m;
            let (ca, cb, cc) = (m3[0].csr(), m3[1].csr(), m3[2].csr());
            let (dx, dw) = (DeviceBuf::from_words(&ctx, words(x))?, DeviceBuf::from_words(&ctx, words(w))?);
            let mut out = [0u64; 32];
            let mut inf = [0u8; 3];
            let rc = unsafe {
                sys::kg_groth16_prove_r1cs_bn254(ctx.raw(), &self.crs[0], &ca, &cb, &cc, dx.as_u64(), dw.as_u64(), r.inner().as_ptr(),
                                                 s.inner().as_ptr(), out.as_mut_ptr(), inf.as_mut_ptr())
            };
            return finish(rc, &out, &inf);
        }
        // several GPUs: the context that runs the transforms evaluates the three matrix-vector products (kg_r1cs_evaluate on
        // the resident CSR), then the task-parallel proof takes the evaluation vectors
        let own = owners(self.n_ctx);
        let (_, guards) = ctxs.lock_set(&own).ok_or(Status(sys::KG_ERR_BAD_ARG))?;      // exactly the contexts the proof uses
        let hc = &*guards[own[2]];
        let shape = self.shape_for(hc, &mut matrices, x.len(), x.len() + w.len())?;
        let m3 = &shape.as_ref().unwrap().m;
        let m = self.crs[0].m;
        let z: Vec<u64> = words(x).iter().chain(words(w).iter()).copied().collect();
        let dz = DeviceBuf::from_words(hc, &z)?;
        let ev = [DeviceBuf::new(hc, 32 * m)?, DeviceBuf::new(hc, 32 * m)?, DeviceBuf::new(hc, 32 * m)?];
        for v in 0..3 {
            let k = m3[v].csr();
            let rc = unsafe { sys::kg_r1cs_evaluate(hc.raw(), k.d_row_ptr, k.d_col, k.d_val, m, dz.as_u64(), ev[v].as_u64()) };
            if rc != sys::KG_OK {
                return Err(Status(rc));
            }
        }
        self.prove_sharded(&guards, [ev[0].as_u64(), ev[1].as_u64(), ev[2].as_u64()], x, w, r, s)
    }

    /// `prove_cs_with` for a caller that holds the matrices anyway (they are only looked at when the resident copy is missing
    /// or being verified).
    #[allow(clippy::too_many_arguments)]
    pub fn prove_cs(&self, a: &SparseMatrix<Fr>, b: &SparseMatrix<Fr>, c: &SparseMatrix<Fr>, x: &[Fr], w: &[Fr], r: &Fr, s: &Fr)
                    -> Result<(G1Affine, G2Affine, G1Affine), Status> {
        // (the content hash of a first sight or a verification: content_hash([a, b, c], x.len()) inside shape_for)
        self.prove_cs_with(|| (a.clone(), b.clone(), c.clone()), x, w, r, s)
    }

    /// the task-parallel proof over `n_ctx` contexts: x, w go to the contexts that run the queries against z, the evaluation
    /// vectors (device pointers on the transform context) stay where they are
    fn prove_sharded(&self, guards: &[std::sync::MutexGuard<'_, Context>], ev: [*mut u64; 3], x: &[Fr], w: &[Fr], r: &Fr, s: &Fr)
                     -> Result<(G1Affine, G2Affine, G1Affine), Status> {
        let n = self.n_ctx;
        let own = owners(n);
        let words = |v: &[Fr]| unsafe { core::slice::from_raw_parts(v.as_ptr() as *const u64, 4 * v.len()) };
        let null = core::ptr::null::<u64>();
        let (mut pa, mut pb, mut pc, mut px, mut pw) = (vec![null; n], vec![null; n], vec![null; n], vec![null; n], vec![null; n]);
        let mut keep = Vec::new();
        for ci in 0..n {
            if ci == own[0] || ci == own[1] {
                let (dx, dw) = (DeviceBuf::from_words(&guards[ci], words(x))?, DeviceBuf::from_words(&guards[ci], words(w))?);
                px[ci] = dx.as_u64() as *const u64;
                pw[ci] = dw.as_u64() as *const u64;
                keep.push(dx);
                keep.push(dw);
            }
        }
        pa[own[2]] = ev[0] as *const u64;
        pb[own[2]] = ev[1] as *const u64;
        pc[own[2]] = ev[2] as *const u64;
        let raws: Vec<*mut sys::KgCtx> = guards[..n].iter().map(|g| g.raw()).collect();
        let crs: Vec<*const sys::KgGroth16Crs> = self.crs.iter().map(|c| c as *const sys::KgGroth16Crs).collect();
        let mut out = [0u64; 32];
        let mut inf = [0u8; 3];
        let rc = unsafe {
            sys::kg_groth16_prove_sharded(raws.as_ptr(), n as i32, crs.as_ptr(), pa.as_ptr(), pb.as_ptr(), pc.as_ptr(), px.as_ptr(), pw.as_ptr(),
                                          r.inner().as_ptr(), s.inner().as_ptr(), out.as_mut_ptr(), inf.as_mut_ptr())
        };
        drop(keep);
        finish(rc, &out, &inf)
    }

    /// One proof: (a, b, c) = cs.evaluate(), x = cs.x(), w = cs.w(), (r, s) drawn by the caller from its rng exactly
    /// as prover.rs:71-72 does.  `Err(Status(KG_ERR_CRS))` is `Error::ProverSubVersionCrsAttack` (prover.rs:67-69).
    pub fn prove(&self, a: &[Fr], b: &[Fr], c: &[Fr], x: &[Fr], w: &[Fr], r: &Fr, s: &Fr) -> Result<(G1Affine, G2Affine, G1Affine), Status> {
        let ctxs = contexts().ok_or(Status(sys::KG_ERR_NO_DEVICE))?;
        let words = |v: &[Fr]| unsafe { core::slice::from_raw_parts(v.as_ptr() as *const u64, 4 * v.len()) };
        if self.n_ctx > 1 {
            let (_, guards) = ctxs.lock_set(&owners(self.n_ctx)).ok_or(Status(sys::KG_ERR_BAD_ARG))?;
            let hc = &*guards[owners(self.n_ctx)[2]];
            let (da, db, dc) = (DeviceBuf::from_words(hc, words(a))?, DeviceBuf::from_words(hc, words(b))?, DeviceBuf::from_words(hc, words(c))?);
            return self.prove_sharded(&guards, [da.as_u64(), db.as_u64(), dc.as_u64()], x, w, r, s);
        }
        let ctx = ctxs.lock(0).ok_or(Status(sys::KG_ERR_BAD_ARG))?;
        let up = |v: &[Fr]| DeviceBuf::from_words(&ctx, words(v));
        let (da, db, dc, dx, dw) = (up(a)?, up(b)?, up(c)?, up(x)?, up(w)?);
        let mut out = [0u64; 32];
        let mut inf = [0u8; 3];
        let rc = unsafe {
            sys::kg_groth16_prove_bn254(ctx.raw(), &self.crs[0], da.as_u64(), db.as_u64(), dc.as_u64(), dx.as_u64(), dw.as_u64(),
                                        r.inner().as_ptr(), s.inner().as_ptr(), out.as_mut_ptr(), inf.as_mut_ptr())
        };
        finish(rc, &out, &inf)
    }
}

fn finish(rc: i32, out: &[u64; 32], inf: &[u8; 3]) -> Result<(G1Affine, G2Affine, G1Affine), Status> {
    if rc != sys::KG_OK {
        return Err(Status(rc));
    }
    Ok((G1Affine::affine_from(&out[0..8], inf[0] != 0), G2Affine::affine_from(&out[8..24], inf[1] != 0),
        G1Affine::affine_from(&out[24..32], inf[2] != 0)))
}

/// The resident CRS of a `Parameters` value, uploaded on first use.  `groth16::Prover { params }` is built by struct
/// literal in zksnark.rs:126, so the device handle is cached here (keyed by the address and length of `params.a`, guarded
/// by its first and last point) instead of in a new field.  The circuit's resident matrices hang off the same entry.
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

/// What `ZkSnark::setup` assembles into `Parameters` / `VerifyingKey` (groth16/src/zksnark.rs:104-124).
pub struct SetupOutput {
    pub h: Vec<G1Affine>,
    pub l: Vec<G1Affine>,
    pub a: Vec<G1Affine>,
    pub b_g1: Vec<G1Affine>,
    pub b_g2: Vec<G2Affine>,
    pub ic: Vec<G1Affine>,
    pub alpha_g1: G1Affine,
    pub beta_g1: G1Affine,
    pub beta_g2: G2Affine,
    pub gamma_g2: G2Affine,
    pub delta_g1: G1Affine,
    pub delta_g2: G2Affine,
}

fn download<C: GpuCurve>(xy: &DeviceBuf, inf: &DeviceBuf, n: usize) -> Result<Vec<C>, Status> {
    let mut w = vec![0u64; n * C::WORDS];
    let mut f = vec![0u64; (n + 7) / 8];
    xy.read_words(&mut w)?;
    inf.read_words(&mut f)?;
    let flags = unsafe { core::slice::from_raw_parts(f.as_ptr() as *const u8, n) };
    Ok((0..n).map(|i| C::affine_from(&w[i * C::WORDS..(i + 1) * C::WORDS], flags[i] != 0)).collect())
}

/// `ZkSnark::setup` after circuit synthesis and after the toxic waste has been drawn (zksnark.rs:28-38): the CRS of the
/// circuit (a, b, c) = cs.matrices() from toxic = [alpha, beta, gamma, delta, tau], computed by `kg_groth16_setup_bn254` on GPU 0
/// -- powers of tau, idft, transposition (x_and_w), the transposed products (eval_at_tau), the linear combinations and every
/// generator multiple.  `None` (no device, any non-zero status -- a zero gamma / delta never reaches this call: the
/// reference has returned `ProverInversionFailed` before) lets the CPU body run.
pub fn setup(a: &SparseMatrix<Fr>, b: &SparseMatrix<Fr>, c: &SparseMatrix<Fr>, m: usize, n_inputs: usize, n_aux: usize, toxic: &[Fr; 5])
             -> Option<SetupOutput> {
    if m == 0 {
        return None;
    }
    let ctxs = contexts()?;
    let ctx = ctxs.lock(0)?;
    let ctx = &*ctx;
    let print = content_hash([a, b, c], n_inputs)?;
    let shape = ResidentShape::build(ctx, a, b, c, n_inputs, print)?;
    let nv = n_inputs + n_aux;
    if !shape.covers(m, nv) {
        return None;
    }
    let buf = |points: usize, words: usize| -> Option<(DeviceBuf, DeviceBuf)> {
        Some((DeviceBuf::new(ctx, points.max(1) * words * 8).ok()?, DeviceBuf::new(ctx, (points.max(1) + 7) / 8 * 8).ok()?))
    };
    let (h, l, qa, qb1, qb2, ic) = (buf(m - 1, 8)?, buf(n_aux, 8)?, buf(nv, 8)?, buf(nv, 8)?, buf(nv, 16)?, buf(n_inputs, 8)?);
    let mut crs = sys::KgGroth16Crs {
        m, l: n_inputs, m_l_1: n_aux,
        d_h: h.0.as_u64(), d_h_inf: h.1.as_u8(), d_l: l.0.as_u64(), d_l_inf: l.1.as_u8(), d_a: qa.0.as_u64(), d_a_inf: qa.1.as_u8(),
        d_b_g1: qb1.0.as_u64(), d_b_g1_inf: qb1.1.as_u8(), d_b_g2: qb2.0.as_u64(), d_b_g2_inf: qb2.1.as_u8(),
        alpha_g1: [0; 8], beta_g1: [0; 8], delta_g1: [0; 8], beta_g2: [0; 16], delta_g2: [0; 16], delta_g1_inf: 0, delta_g2_inf: 0,
    };
    let (ca, cb, cc) = (shape.m[0].csr(), shape.m[1].csr(), shape.m[2].csr());
    let mut gamma_g2 = [0u64; 16];
    let mut vk_inf = [0u8; 6];
    let rc = unsafe {
        sys::kg_groth16_setup_bn254(ctx.raw(), &ca, &cb, &cc, m, n_inputs, n_aux, toxic.as_ptr() as *const u64, &mut crs, ic.0.as_u64(),
                                    ic.1.as_u8(), gamma_g2.as_mut_ptr(), vk_inf.as_mut_ptr())
    };
    if rc != sys::KG_OK {
        return None;
    }
    Some(SetupOutput {
        h: download(&h.0, &h.1, m - 1).ok()?,
        l: download(&l.0, &l.1, n_aux).ok()?,
        a: download(&qa.0, &qa.1, nv).ok()?,
        b_g1: download(&qb1.0, &qb1.1, nv).ok()?,
        b_g2: download(&qb2.0, &qb2.1, nv).ok()?,
        ic: download(&ic.0, &ic.1, n_inputs).ok()?,
        alpha_g1: G1Affine::affine_from(&crs.alpha_g1, vk_inf[0] != 0),
        beta_g1: G1Affine::affine_from(&crs.beta_g1, vk_inf[1] != 0),
        delta_g1: G1Affine::affine_from(&crs.delta_g1, vk_inf[2] != 0),
        beta_g2: G2Affine::affine_from(&crs.beta_g2, vk_inf[3] != 0),
        gamma_g2: G2Affine::affine_from(&gamma_g2, vk_inf[4] != 0),
        delta_g2: G2Affine::affine_from(&crs.delta_g2, vk_inf[5] != 0),
    })
}

impl Drop for ResidentCrs {
    fn drop(&mut self) {
        if let Some(ctxs) = contexts() {
            for (ci, p) in self.registered.iter() {
                if let Some(c) = ctxs.lock(*ci) {
                    unsafe { sys::kg_bases_unregister(c.raw(), *p) };
                }
            }
        }
    }
}

//! `nova/src/pedersen.rs:10-20` on the devices of the node: the commitment key `g` is uploaded once, slice i to GPU i
//! (`kg_sharded_key_create`: index-range sharding), and stays resident in the MSM's internal form;
//! `commit` uploads each device's slice of `m`, every device runs the full pipeline on its slice and the per-device
//! affine partial sums are added on the host (`kg_sharded_key_commit`).  The reference re-reads `g` on every call and
//! folds naive double-and-add scalar multiplications.
use std::collections::HashMap;
use std::sync::Mutex;

use kogarashi_amd_sys as sys;
use zkstd::common::BNAffine;

use crate::{cast_slice, contexts, marshal, scalar_words, GpuCurve};
use bn_254::G1Affine;
use grumpkin::Affine as GkAffine;

struct Resident {
    key: *mut sys::KgShardedKey,
    fingerprint: [u64; 6],
}
unsafe impl Send for Resident {}

/// keys by (address of g, length): `PedersenCommitment` derives Clone / Encode / Decode / PartialEq, so the device handle
/// lives here instead of in the struct; the fingerprint (first, middle and last generator) guards against a freed and
/// re-used allocation.  This is an ADDRESS-keyed cache with a three-point probe, not a content hash: a key whose interior
/// generators are overwritten in place while those three stay is not noticed -- `PedersenCommitment { g }` is immutable in the
/// reference (`nova/src/pedersen.rs:6-13`: built by `new`, only read afterwards), which is what makes the probe sufficient.
static KEYS: Mutex<Option<HashMap<(usize, usize, i32), Resident>>> = Mutex::new(None);     // const `Mutex::new`: no `OnceLock` on the pinned nightly

fn fingerprint(xy: &[u64], words: usize, n: usize) -> [u64; 6] {
    let at = |i: usize| (xy[i * words], xy[i * words + words / 2]);
    let (a, b, c) = (at(0), at(n / 2), at(n - 1));
    [a.0, a.1, b.0, b.1, c.0, c.1]
}

fn commit_typed<C: GpuCurve>(g: &[C], m: *const u64, m_len: usize) -> Option<C> {
    if g.is_empty() {
        return None;
    }
    let ctxs = contexts()?;
    let all = ctxs.lock_all()?;                                // the key is spread over every GPU of the node: the call holds them all
    let mut keys_guard = KEYS.lock().ok()?;
    let keys = keys_guard.get_or_insert_with(HashMap::new);
    let id = (g.as_ptr() as usize, g.len(), C::CURVE);
    // the probe marshals three points only; the whole key is marshalled when it is first seen
    let probe = {
        let pts = [g[0], g[g.len() / 2], g[g.len() - 1]];
        let (xy, _) = marshal(&pts);
        fingerprint(&xy, C::WORDS, 3)
    };
    let stale = keys.get(&id).map(|r| r.fingerprint != probe).unwrap_or(false);
    if stale {
        if let Some(r) = keys.remove(&id) {
            unsafe { sys::kg_sharded_key_destroy(r.key) };
        }
    }
    if !keys.contains_key(&id) {
        let (xy, inf) = marshal(g);
        let raws: Vec<*mut sys::KgCtx> = all.iter().map(|c| c.raw()).collect();
        let mut key = core::ptr::null_mut();
        let rc = unsafe {
            sys::kg_sharded_key_create(raws.as_ptr(), raws.len() as i32, C::CURVE, xy.as_ptr(), inf.as_ptr(), g.len(), &mut key)
        };
        if rc != sys::KG_OK {
            return None;
        }
        keys.insert(id, Resident { key, fingerprint: probe });
    }
    let r = keys.get(&id)?;
    let mut xy = [0u64; 16];
    let mut inf = 0u8;
    let rc = unsafe { sys::kg_sharded_key_commit(r.key, m, m_len, xy.as_mut_ptr(), &mut inf) };
    if rc != sys::KG_OK {
        return None;
    }
    Some(C::affine_from(&xy, inf != 0))
}

/// `PedersenCommitment::commit`: affine(sum_i g[i] * m[i]) over min(len) pairs.
pub fn commit<C: BNAffine + 'static>(g: &[C], m: &[C::Scalar]) -> Option<C>
where
    C::Scalar: 'static,
{
    let (mw, _) = scalar_words(m)?;
    if let Some(b) = cast_slice::<C, G1Affine>(g) {
        return crate::cast_val(commit_typed(b, mw, m.len())?);
    }
    if let Some(b) = cast_slice::<C, GkAffine>(g) {
        return crate::cast_val(commit_typed(b, mw, m.len())?);
    }
    None
}

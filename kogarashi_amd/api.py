"""Host-side mirror of the reference's call sites for the hot path (same names, argument meaning and error
behaviour), implemented on top of the C ABI.  Field elements are numpy uint64 arrays of 4 little-endian
Montgomery limbs (the reference's `Fr([u64; 4])` / `Fq`), points are (x | y) rows plus an infinity-flag array.

  msm_curve_addition(bases, coeffs)            groth16/src/msm.rs:6-48
  Fft(k).dft / idft / coset_dft / coset_idft / divide_by_z_on_coset      groth16/src/fft.rs:27-154
  PedersenCommitment(g).commit(m)              nova/src/pedersen.rs:10-20
  Prover(params).create_proof(...)             groth16/src/prover.rs:14-99
"""
from __future__ import annotations

import numpy as np

from .lib import KG_FQ, KG_FR, KG_G1, KG_G2, KG_GRUMPKIN, Context, Groth16Crs

_default_ctx = None
CURVE_IDS = {"g1": KG_G1, "grumpkin": KG_GRUMPKIN, "gk": KG_GRUMPKIN, "g2": KG_G2}
_LIMBS = {KG_G1: 8, KG_GRUMPKIN: 8, KG_G2: 16}


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def msm_curve_addition(bases, coeffs, curve="g1", is_infinity=None, ctx: Context | None = None) -> np.ndarray:
    """Variable-base MSM over min(len(bases), len(coeffs)) pairs (the reference zips, msm.rs:25).
    Returns the projective sum as 3 base-field elements (x, y, z), normalised to z = 1 / (0, 1, 0)."""
    ctx = ctx or default_context()
    cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
    w = _LIMBS[cid]
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    n = min(len(bases), len(coeffs))
    inf = None
    if is_infinity is not None:
        inf = np.ascontiguousarray(is_infinity, dtype=np.uint8)[:n]
    return ctx.msm_host(cid, bases[:n], inf, coeffs[:n], n)


class Fft:
    """groth16::fft::Fft<Fr>: size-2^k transforms, natural order in and out; inputs shorter than n are
    zero-padded like prepare_fft (fft.rs:157-162).  Unlike the reference, twiddles are cached on the device."""

    def __init__(self, k: int, ctx: Context | None = None):
        if k < 1:
            raise AssertionError("k >= 1")          # fft.rs:28 assert!(k >= 1)
        if k > 28:
            raise ValueError("k exceeds the two-adicity S = 28 of Fr")
        self.k, self.n = k, 1 << k
        self.ctx = ctx or default_context()

    def _run(self, v, inverse, coset):
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        buf = np.zeros((self.n, 4), dtype=np.uint64)
        buf[: min(len(v), self.n)] = v[: self.n]
        d = self.ctx.upload(buf)
        self.ctx.ntt(d.ptr, self.k, inverse, coset)
        return d.numpy()

    def dft(self, coeffs):
        return self._run(coeffs, False, False)

    def idft(self, points):
        return self._run(points, True, False)

    def coset_dft(self, coeffs):
        return self._run(coeffs, False, True)

    def coset_idft(self, points):
        return self._run(points, True, True)

    def divide_by_z_on_coset(self, points):
        v = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 4)
        buf = np.zeros((self.n, 4), dtype=np.uint64)
        buf[: min(len(v), self.n)] = v[: self.n]
        d = self.ctx.upload(buf)
        self.ctx.divide_by_z_on_coset(d.ptr, self.k)
        return d.numpy()


class PedersenCommitment:
    """nova::PedersenCommitment<C>: commit(m) = affine(sum_i g[i] * m[i]) over min(len) pairs.
    The generators are uploaded once and stay resident (the reference re-reads them on every call)."""

    def __init__(self, g, curve="g1", is_infinity=None, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        w = _LIMBS[self.cid]
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, w)
        self.len = len(g)
        self._g = self.ctx.upload(g)
        self._inf = None
        if is_infinity is not None:
            self._inf = self.ctx.upload(np.ascontiguousarray(is_infinity, dtype=np.uint8))

    def commit(self, m):
        m = np.ascontiguousarray(m, dtype=np.uint64).reshape(-1, 4)
        n = min(len(m), self.len)
        d = self.ctx.upload(m[:n])
        return self.ctx.commit(self.cid, self._g.ptr, self._inf.ptr if self._inf else 0, d.ptr, n)


class Prover:
    """groth16::Prover { params }: the CRS (groth16/src/params.rs:6-28) is uploaded once and stays resident.

    params: dict with "h", "l", "a", "b_g1" (G1 rows x|y), "b_g2" (G2 rows), optional "<name>_inf" flag arrays,
    "vk_g1" = rows alpha_g1, beta_g1, delta_g1 and "vk_g2" = rows beta_g2, delta_g2 (affine), optional
    "delta_g1_inf"/"delta_g2_inf"."""

    def __init__(self, params: dict, m: int, l: int, m_l_1: int, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.m, self.l, self.m_l_1 = m, l, m_l_1
        self._keep = []
        crs = Groth16Crs()
        crs.m, crs.l, crs.m_l_1 = m, l, m_l_1
        for name, w in (("h", 8), ("l", 8), ("a", 8), ("b_g1", 8), ("b_g2", 16)):
            arr = np.ascontiguousarray(params[name], dtype=np.uint64).reshape(-1, w)
            d = self.ctx.upload(arr)
            self._keep.append(d)
            setattr(crs, "d_" + name, d.ptr)
            inf = params.get(name + "_inf")
            if inf is not None and np.any(inf):
                di = self.ctx.upload(np.ascontiguousarray(inf, dtype=np.uint8))
                self._keep.append(di)
                setattr(crs, "d_" + name + "_inf", di.ptr)
        g1 = np.ascontiguousarray(params["vk_g1"], dtype=np.uint64).reshape(-1, 8)
        g2 = np.ascontiguousarray(params["vk_g2"], dtype=np.uint64).reshape(-1, 16)
        for i in range(8):
            crs.alpha_g1[i], crs.beta_g1[i], crs.delta_g1[i] = int(g1[0, i]), int(g1[1, i]), int(g1[2, i])
        for i in range(16):
            crs.beta_g2[i], crs.delta_g2[i] = int(g2[0, i]), int(g2[1, i])
        crs.delta_g1_inf = int(bool(params.get("delta_g1_inf", 0)))
        crs.delta_g2_inf = int(bool(params.get("delta_g2_inf", 0)))
        self.crs = crs

    def create_proof(self, a_eval, b_eval, c_eval, x, w, r, s):
        """Proof {a, b, c} for the synthesised constraint system: (A, B, C) = cs.evaluate(), x = cs.x(), w = cs.w();
        (r, s) are the blinding scalars the reference draws from its rng (prover.rs:71-72).
        Raises ProverSubVersionCrsAttack like prover.rs:67-69."""
        up = lambda v: self.ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        da, db, dc, dx, dw = up(a_eval), up(b_eval), up(c_eval), up(x), up(w)
        return self.ctx.groth16_prove(self.crs, da.ptr, db.ptr, dc.ptr, dx.ptr, dw.ptr, r, s)

"""Host-side mirror of the reference's call sites for the hot path (same names, argument meaning and error
behaviour), implemented on top of the C ABI.  Field elements are numpy uint64 arrays of 4 little-endian
Montgomery limbs (the reference's `Fr([u64; 4])` / `Fq`), points are (x | y) rows plus an infinity-flag array.

  msm_curve_addition(bases, coeffs)            groth16/src/msm.rs:6-48
  Fft(k).dft / idft / coset_dft / coset_idft / divide_by_z_on_coset      groth16/src/fft.rs:27-154
  PedersenCommitment(g).commit(m)              nova/src/pedersen.rs:10-20
  Prover(params).create_proof(...)             groth16/src/prover.rs:14-99
  NovaProver(shape, ck).compute_cross_term / commit_t      nova/src/prover.rs:24-90
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
        self.ctx.bases_register(self.cid, self._g.ptr, self._inf.ptr if self._inf else 0, self.len)

    def __del__(self):
        try:
            self.ctx.bases_unregister(self._g.ptr)
        except Exception:
            pass

    def commit(self, m):
        m = np.ascontiguousarray(m, dtype=np.uint64).reshape(-1, 4)
        n = min(len(m), self.len)
        # the key is resident, m is a host slice: uploaded in index slices under the accumulations (kg_commit_host_scalars)
        return self.ctx.commit_host_scalars(self.cid, self._g.ptr, self._inf.ptr if self._inf else 0, m[:n], n)


class Prover:
    """groth16::Prover { params }: the CRS (groth16/src/params.rs:6-28) is uploaded once and stays resident.

    params: dict with "h", "l", "a", "b_g1" (G1 rows x|y), "b_g2" (G2 rows), optional "<name>_inf" flag arrays,
    "vk_g1" = rows alpha_g1, beta_g1, delta_g1 and "vk_g2" = rows beta_g2, delta_g2 (affine), optional
    "delta_g1_inf"/"delta_g2_inf".

    window_tables=True additionally stores every vector's window multiples (kg_bases_precompute: ceil(255 / 17) = 15 x the
    CRS in device memory, built once): the five MSMs of a proof then use one bucket set each.  Offered for circuits whose
    witness and h vectors hold 2^16 .. 2^20 entries; other sizes keep the plain resident form."""

    def __init__(self, params: dict, m: int, l: int, m_l_1: int, ctx: Context | None = None, window_tables: bool = False):
        self.ctx = ctx or default_context()
        self.m, self.l, self.m_l_1 = m, l, m_l_1
        self._keep = []
        self._registered = []
        crs = Groth16Crs()
        crs.m, crs.l, crs.m_l_1 = m, l, m_l_1
        for name, w in (("h", 8), ("l", 8), ("a", 8), ("b_g1", 8), ("b_g2", 16)):
            arr = np.ascontiguousarray(params[name], dtype=np.uint64).reshape(-1, w)
            d = self.ctx.upload(arr)
            self._keep.append(d)
            setattr(crs, "d_" + name, d.ptr)
            inf = params.get(name + "_inf")
            di = None
            if inf is not None and np.any(inf):
                di = self.ctx.upload(np.ascontiguousarray(inf, dtype=np.uint8))
                self._keep.append(di)
                setattr(crs, "d_" + name + "_inf", di.ptr)
            # Parameters is immutable: convert each CRS vector to the device's internal form once
            self.ctx.bases_register(KG_G2 if w == 16 else KG_G1, d.ptr, di.ptr if di else 0, len(arr))
            self._registered.append(d.ptr)
        g1 = np.ascontiguousarray(params["vk_g1"], dtype=np.uint64).reshape(-1, 8)
        g2 = np.ascontiguousarray(params["vk_g2"], dtype=np.uint64).reshape(-1, 16)
        for i in range(8):
            crs.alpha_g1[i], crs.beta_g1[i], crs.delta_g1[i] = int(g1[0, i]), int(g1[1, i]), int(g1[2, i])
        for i in range(16):
            crs.beta_g2[i], crs.delta_g2[i] = int(g2[0, i]), int(g2[1, i])
        crs.delta_g1_inf = int(bool(params.get("delta_g1_inf", 0)))
        crs.delta_g2_inf = int(bool(params.get("delta_g2_inf", 0)))
        self.crs = crs
        self.window_tables = False
        nz, hn = l + m_l_1, m - 1
        if window_tables and all((1 << 16) <= v <= (1 << 20) for v in (nz, hn)) and len(params["h"]) == hn:
            for name in ("a", "b_g1", "b_g2", "l"):
                self.ctx.bases_precompute(getattr(crs, "d_" + name), nz)      # l meets the whole witness vector z = x || w
            self.ctx.bases_precompute(crs.d_h, hn)
            self.window_tables = True

    def __del__(self):
        for p_ in getattr(self, "_registered", []):
            try:
                self.ctx.bases_unregister(p_)
            except Exception:
                pass

    def create_proof(self, a_eval, b_eval, c_eval, x, w, r, s):
        """Proof {a, b, c} for the synthesised constraint system: (A, B, C) = cs.evaluate(), x = cs.x(), w = cs.w();
        (r, s) are the blinding scalars the reference draws from its rng (prover.rs:71-72).
        Raises ProverSubVersionCrsAttack like prover.rs:67-69."""
        up = lambda v: self.ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        da, db, dc, dx, dw = up(a_eval), up(b_eval), up(c_eval), up(x), up(w)
        return self.ctx.groth16_prove(self.crs, da.ptr, db.ptr, dc.ptr, dx.ptr, dw.ptr, r, s)

    def attach_constraint_system(self, a, b, c):
        """The circuit's constraint matrices (CSR triples (row_ptr, col, val) over z = x || w, cs.matrices()): uploaded once;
        create_proof_from_witness then runs cs.evaluate() (zkstd/src/r1cs.rs:137-142) on the device as well."""
        self._cs = []
        for rp, col, val in (a, b, c):
            val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4)
            col = np.ascontiguousarray(col, dtype=np.uint64)
            self._cs.append((self.ctx.upload(np.ascontiguousarray(rp, dtype=np.uint64)),
                             self.ctx.upload(col if len(col) else np.zeros(1, dtype=np.uint64)),
                             self.ctx.upload(val if len(val) else np.zeros((1, 4), dtype=np.uint64))))

    def create_proof_from_witness(self, x, w, r, s):
        """create_proof from (x, w) alone: the evaluation vectors are made on the device (kg_groth16_prove_r1cs_bn254)"""
        up = lambda v: self.ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        dx, dw = up(x), up(w)
        ptrs = [tuple(d.ptr for d in trip) for trip in self._cs]
        return self.ctx.groth16_prove_r1cs(self.crs, ptrs[0], ptrs[1], ptrs[2], dx.ptr, dw.ptr, r, s)

    def create_proofs(self, jobs):
        """Proofs for an iterable of (a_eval, b_eval, c_eval, x, w, r, s), two in flight (kg_groth16_prove_begin / _end):
        proof i+1 is enqueued before proof i is collected, so its transforms and sorts overlap proof i's last reduction
        and host assembly.  Yields the proofs in order."""
        up = lambda v: self.ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        held = [None, None]
        in_flight = []                                  # tickets begun and not yet ended, oldest first
        try:
            for i, (a_eval, b_eval, c_eval, x, w, r, s) in enumerate(jobs):
                t = i & 1
                held[t] = [up(a_eval), up(b_eval), up(c_eval), up(x), up(w)]       # alive until the matching end
                da, db, dc, dx, dw = held[t]
                self.ctx.groth16_prove_begin(self.crs, da.ptr, db.ptr, dc.ptr, dx.ptr, dw.ptr, r, s, t)
                in_flight.append(t)
                if len(in_flight) == 2:
                    yield self.ctx.groth16_prove_end(in_flight.pop(0))
            while in_flight:
                yield self.ctx.groth16_prove_end(in_flight.pop(0))
        finally:
            for t in in_flight:                         # an error or an abandoned generator: every begin gets its end
                try:
                    self.ctx.groth16_prove_end(t)
                except Exception:
                    pass


class ShardedProver:
    """groth16::Prover over several contexts (normally one per GPU), task-parallel: SURVEY.md 8e -- the MSMs of
    prover.rs:51-65 are independent until the assembly and the G2 query is the long pole.  Context 0 holds b_g2 and runs
    that query, context 1 % n holds a, b_g1, l (the three G1 queries against z = x || w), context 2 % n holds h and runs the
    transforms and h's MSM (kg_groth16_prove_sharded); each vector is uploaded and registered only where it is used."""

    ROLE_VECTORS = (("b_g2",), ("a", "b_g1", "l"), ("h",))

    def __init__(self, params: dict, m: int, l: int, m_l_1: int, ctxs):
        from .lib import groth16_prove_sharded
        self._prove = groth16_prove_sharded
        self.ctxs = list(ctxs)
        n = len(self.ctxs)
        self.owner = [0, 1 % n, 2 % n]
        self.m, self.l, self.m_l_1 = m, l, m_l_1
        self._keep, self._registered = [], []
        g1 = np.ascontiguousarray(params["vk_g1"], dtype=np.uint64).reshape(-1, 8)
        g2 = np.ascontiguousarray(params["vk_g2"], dtype=np.uint64).reshape(-1, 16)
        self.crs = []
        for ci, ctx in enumerate(self.ctxs):
            crs = Groth16Crs()
            crs.m, crs.l, crs.m_l_1 = m, l, m_l_1
            for role, names in enumerate(self.ROLE_VECTORS):
                if self.owner[role] != ci:
                    continue
                for name in names:
                    w = 16 if name == "b_g2" else 8
                    arr = np.ascontiguousarray(params[name], dtype=np.uint64).reshape(-1, w)
                    d = ctx.upload(arr)
                    self._keep.append(d)
                    setattr(crs, "d_" + name, d.ptr)
                    inf = params.get(name + "_inf")
                    di = None
                    if inf is not None and np.any(inf):
                        di = ctx.upload(np.ascontiguousarray(inf, dtype=np.uint8))
                        self._keep.append(di)
                        setattr(crs, "d_" + name + "_inf", di.ptr)
                    ctx.bases_register(KG_G2 if w == 16 else KG_G1, d.ptr, di.ptr if di else 0, len(arr))
                    self._registered.append((ctx, d.ptr))
            for i in range(8):
                crs.alpha_g1[i], crs.beta_g1[i], crs.delta_g1[i] = int(g1[0, i]), int(g1[1, i]), int(g1[2, i])
            for i in range(16):
                crs.beta_g2[i], crs.delta_g2[i] = int(g2[0, i]), int(g2[1, i])
            crs.delta_g1_inf = int(bool(params.get("delta_g1_inf", 0)))
            crs.delta_g2_inf = int(bool(params.get("delta_g2_inf", 0)))
            self.crs.append(crs)

    def __del__(self):
        for ctx, p_ in getattr(self, "_registered", []):
            try:
                ctx.bases_unregister(p_)
            except Exception:
                pass

    def create_proof(self, a_eval, b_eval, c_eval, x, w, r, s):
        """as Prover.create_proof; the witness goes to the contexts that run the queries against z, the evaluation vectors to
        the one that runs the transforms"""
        n = len(self.ctxs)
        up = lambda ctx, v: ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        keep, ptr = [], {k: [0] * n for k in "abcxw"}
        for ci, ctx in enumerate(self.ctxs):
            if ci in self.owner[:2]:
                for k, v in (("x", x), ("w", w)):
                    d = up(ctx, v)
                    keep.append(d)
                    ptr[k][ci] = d.ptr
            if ci == self.owner[2]:
                for k, v in (("a", a_eval), ("b", b_eval), ("c", c_eval)):
                    d = up(ctx, v)
                    keep.append(d)
                    ptr[k][ci] = d.ptr
        return self._prove(self.ctxs, self.crs, ptr["a"], ptr["b"], ptr["c"], ptr["x"], ptr["w"], r, s)

    def upload_inputs(self, a_eval, b_eval, c_eval, x, w):
        """the inputs of create_proof made resident where they are read (returns the per-context pointer lists and the arrays
        that keep them alive): repeated proofs over the same inputs then move nothing over PCIe (bench.py groth16.sharded)"""
        n = len(self.ctxs)
        up = lambda ctx, v: ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        keep, ptr = [], {k: [0] * n for k in "abcxw"}
        for ci, ctx in enumerate(self.ctxs):
            if ci in self.owner[:2]:
                for k, v in (("x", x), ("w", w)):
                    d = up(ctx, v)
                    keep.append(d)
                    ptr[k][ci] = d.ptr
            if ci == self.owner[2]:
                for k, v in (("a", a_eval), ("b", b_eval), ("c", c_eval)):
                    d = up(ctx, v)
                    keep.append(d)
                    ptr[k][ci] = d.ptr
        return ptr, keep

    def prove_resident(self, ptr, r, s):
        return self._prove(self.ctxs, self.crs, ptr["a"], ptr["b"], ptr["c"], ptr["x"], ptr["w"], r, s)



class NovaProver:
    """nova::Prover { ck, shape } as far as the hot path goes: the cross term T of a folding step and its commitment
    (nova/src/prover.rs:31-35, 53-90).  shape: the R1CS matrices (a, b, c) as CSR triples (row_ptr, col, val) over
    z = (u | x | w) -- column 0 is the relaxed one-wire, instance wire i is column i, witness wire k is column l + k with
    l = len(x) + 1 (SparseMatrix::prod's index rule, zkstd/src/matrix.rs:36-48).  Scalars are Fr for the bn254 driver and
    Fq for the Grumpkin driver (nova/src/driver.rs:9-42).  The matrices are uploaded once and stay resident."""

    def __init__(self, shape, ck: "PedersenCommitment", ctx: Context | None = None):
        self.ctx = ctx or ck.ctx
        self.ck = ck
        self.field = KG_FQ if ck.cid == KG_GRUMPKIN else KG_FR
        self.m = len(shape[0][0]) - 1
        self._dev = []
        for rp, col, val in shape:
            val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4)
            col = np.ascontiguousarray(col, dtype=np.uint64)
            self._dev.append((self.ctx.upload(np.ascontiguousarray(rp, dtype=np.uint64)),
                              self.ctx.upload(col if len(col) else np.zeros(1, dtype=np.uint64)),
                              self.ctx.upload(val if len(val) else np.zeros((1, 4), dtype=np.uint64))))

    def _z(self, u, x, w):
        parts = [np.ascontiguousarray(u, dtype=np.uint64).reshape(1, 4), np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4),
                 np.ascontiguousarray(w, dtype=np.uint64).reshape(-1, 4)]
        return self.ctx.upload(np.concatenate(parts))

    def compute_cross_term_device(self, u1, x1, w1, u2, x2, w2):
        """T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1 as a device array of m elements (prover.rs:53-90)"""
        z1, z2 = self._z(u1, x1, w1), self._z(u2, x2, w2)
        t = self.ctx.empty((self.m, 4))
        ptrs = [tuple(d.ptr for d in trip) for trip in self._dev]
        self.ctx.nova_cross_term(self.field, ptrs[0], ptrs[1], ptrs[2], self.m, z1.ptr, z2.ptr, u1, u2, t.ptr)
        return t

    def compute_cross_term(self, u1, x1, w1, u2, x2, w2) -> np.ndarray:
        return self.compute_cross_term_device(u1, x1, w1, u2, x2, w2).numpy()

    def _one(self):
        """the scalar field's one in Montgomery form, made by the device (to_mont of the integer 1)"""
        if getattr(self, "_one_cache", None) is None:
            d = self.ctx.upload(np.array([[1, 0, 0, 0]], dtype=np.uint64))
            self.ctx.field_vec_op(self.field, "to_mont", d.ptr, 0, d.ptr, 1)
            self._one_cache = d.numpy()[0].copy()
        return self._one_cache

    def prove(self, instance1: dict, witness1: dict, instance2: dict, witness2: dict, r):
        """nova::Prover::prove (nova/src/prover.rs:24-50) with the challenge r supplied by the caller -- the transcript hash
        that produces it is sequential host work outside the hot path.

        instance1: relaxed instance {"commit_w": (xy, inf), "commit_e": (xy, inf), "u": limbs, "x": (k, 4)};
        witness1: {"w": (n, 4), "e": (m, 4)}; instance2: {"commit_w": (xy, inf), "x": (k, 4)}; witness2: {"w": (n, 4)}.
        Returns (instance, witness, commit_t) in the same shapes.  On the device: the cross term, its commitment and both
        witness folds (W1 + r W2, E1 + r T: relaxed_r1cs/witness.rs:56-70); the instance fold (instance.rs:81-101) is two
        two-point MSMs and a handful of scalars."""
        c, fd, cid = self.ctx, self.field, self.ck.cid
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        one = self._one()
        u1 = np.ascontiguousarray(instance1["u"], dtype=np.uint64).reshape(4)
        x1 = np.ascontiguousarray(instance1["x"], dtype=np.uint64).reshape(-1, 4)
        x2 = np.ascontiguousarray(instance2["x"], dtype=np.uint64).reshape(-1, 4)
        w1 = np.ascontiguousarray(witness1["w"], dtype=np.uint64).reshape(-1, 4)
        w2 = np.ascontiguousarray(witness2["w"], dtype=np.uint64).reshape(-1, 4)
        t = self.compute_cross_term_device(u1, x1, w1, one, x2, w2)
        n = min(self.m, self.ck.len)
        commit_t = c.commit(cid, self.ck._g.ptr, self.ck._inf.ptr if self.ck._inf else 0, t.ptr, n)
        # witness fold
        dw1, dw2, de1 = c.upload(w1), c.upload(w2), c.upload(np.ascontiguousarray(witness1["e"], dtype=np.uint64).reshape(-1, 4))
        c.field_vec_axpy(fd, dw1.ptr, r, dw2.ptr, dw1.ptr, len(w1))
        c.field_vec_axpy(fd, de1.ptr, r, t.ptr, de1.ptr, self.m)
        # instance fold: u = u1 + r, x = x1 + r x2 (one small axpy over u | x against 1 | x2)
        ux1, ux2 = c.upload(np.concatenate([u1.reshape(1, 4), x1])), c.upload(np.concatenate([one.reshape(1, 4), x2]))
        c.field_vec_axpy(fd, ux1.ptr, r, ux2.ptr, ux1.ptr, 1 + len(x1))
        ux = ux1.numpy()

        def fold_point(p1, p2):                     # p1 + r * p2 as an affine point: msm([p1, p2], [1, r])
            w = 16 if cid == KG_G2 else 8
            pts = np.stack([np.ascontiguousarray(p1[0], dtype=np.uint64).reshape(w), np.ascontiguousarray(p2[0], dtype=np.uint64).reshape(w)])
            flags = np.array([int(bool(p1[1])), int(bool(p2[1]))], dtype=np.uint8)
            out = c.msm_host(cid, pts, flags, np.stack([one, r]), 2)
            return out[:w].copy(), int(not out[w:].any())

        instance = {"commit_w": fold_point(instance1["commit_w"], instance2["commit_w"]),
                    "commit_e": fold_point(instance1["commit_e"], commit_t), "u": ux[0].copy(), "x": ux[1:].copy()}
        return instance, {"w": dw1.numpy(), "e": de1.numpy()}, commit_t

    def commit_t(self, u1, x1, w1, u2, x2, w2):
        """(T, commit_T): the cross term and ck.commit(&t) (prover.rs:33-35); T never leaves the device in between"""
        t = self.compute_cross_term_device(u1, x1, w1, u2, x2, w2)
        n = min(self.m, self.ck.len)
        xy, inf = self.ctx.commit(self.ck.cid, self.ck._g.ptr, self.ck._inf.ptr if self.ck._inf else 0, t.ptr, n)
        return t.numpy(), (xy, inf)


def groth16_setup(a, b, c, m: int, l: int, m_l_1: int, toxic, field_ops=None, ctx: Context | None = None) -> dict:
    """ZkSnark::setup (groth16/src/zksnark.rs:17-127) on the device through kg_groth16_setup_bn254, with the toxic waste injected
    (alpha, beta, gamma, delta, tau as rows of `toxic`; the reference draws them from its rng, :28-32).

    a, b, c: CSR triples (row_ptr, col, val) over z = x || w, host arrays.  Returns the Parameters dict `Prover` takes
    (device-computed, downloaded).  `field_ops` is accepted for callers of the round-3 signature and ignored: the five host-side
    scalars are computed behind the boundary now."""
    ctx = ctx or default_context()
    toxic = np.ascontiguousarray(toxic, dtype=np.uint64).reshape(5, 4)
    nv = l + m_l_1
    keep, mats = [], []
    for rp, col, val in (a, b, c):
        col = np.ascontiguousarray(col, dtype=np.uint64)
        val = np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4)
        d = [ctx.upload(np.ascontiguousarray(rp, dtype=np.uint64)), ctx.upload(col if len(col) else np.zeros(1, dtype=np.uint64)),
             ctx.upload(val if len(val) else np.zeros((1, 4), dtype=np.uint64))]
        keep.append(d)
        mats.append((d[0].ptr, d[1].ptr, d[2].ptr))
    lens = {"h": (max(m - 1, 0), 8), "l": (m_l_1, 8), "a": (nv, 8), "b_g1": (nv, 8), "b_g2": (nv, 16), "ic": (l, 8)}
    dev = {k: (ctx.empty((max(cnt, 1), w)), ctx.empty((max(cnt, 1),), dtype=np.uint8)) for k, (cnt, w) in lens.items()}
    crs = Groth16Crs()
    for k in ("h", "l", "a", "b_g1", "b_g2"):
        setattr(crs, "d_" + k, dev[k][0].ptr)
        setattr(crs, "d_" + k + "_inf", dev[k][1].ptr)
    gamma_g2, vk_inf = ctx.groth16_setup(mats[0], mats[1], mats[2], m, l, m_l_1, toxic, crs, dev["ic"][0].ptr, dev["ic"][1].ptr)
    P = {}
    for k, (cnt, w) in lens.items():
        P[k], P[k + "_inf"] = dev[k][0].numpy()[:cnt], dev[k][1].numpy()[:cnt]
    P["vk_g1"] = np.stack([np.array(crs.alpha_g1, dtype=np.uint64), np.array(crs.beta_g1, dtype=np.uint64), np.array(crs.delta_g1, dtype=np.uint64)])
    P["vk_g2"] = np.stack([np.array(crs.beta_g2, dtype=np.uint64), np.array(crs.delta_g2, dtype=np.uint64)])
    P["gamma_g2"] = gamma_g2
    P["delta_g1_inf"], P["delta_g2_inf"] = int(crs.delta_g1_inf), int(crs.delta_g2_inf)
    return P

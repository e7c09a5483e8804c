"""ctypes binding of liboracle.so (the C restatement of the reference's CPU path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
All field elements are numpy uint64 arrays of 4 little-endian limbs in Montgomery form (R = 2^256),
i.e. the reference's in-memory representation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FR, FQ = 0, 1
U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("kg_oracle.c", "kg_oracle_curve.inc", "kg_oracle_groth16.inc", "kg_oracle_nova.inc")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.kgo_fft_new.restype = C.c_void_p
    return _LIB


def _p(a):
    return a.ctypes.data_as(U64P)


def _u8(a):
    return None if a is None else a.ctypes.data_as(U8P)


def _arr(x, shape=None):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    return a if shape is None else a.reshape(shape)


# ---- field ------------------------------------------------------------------------------------
def _bin(name):
    def f(fd, a, b):
        a, b = _arr(a), _arr(b)
        o = np.empty(4, dtype=np.uint64)
        getattr(lib(), name)(fd, _p(a), _p(b), _p(o))
        return o
    return f


def _un(name):
    def f(fd, a):
        a = _arr(a)
        o = np.empty(4, dtype=np.uint64)
        getattr(lib(), name)(fd, _p(a), _p(o))
        return o
    return f


f_add, f_sub, f_mul = _bin("kgo_f_add"), _bin("kgo_f_sub"), _bin("kgo_f_mul")
f_double, f_neg, f_square = _un("kgo_f_double"), _un("kgo_f_neg"), _un("kgo_f_square")
f_to_mont, f_from_mont = _un("kgo_f_to_mont"), _un("kgo_f_from_mont")


def f_mont(fd, a8):
    a8 = _arr(a8)
    o = np.empty(4, dtype=np.uint64)
    lib().kgo_f_mont(fd, _p(a8), _p(o))
    return o


def f_from_u512(fd, a8):
    a8 = _arr(a8)
    o = np.empty(4, dtype=np.uint64)
    lib().kgo_f_from_u512(fd, _p(a8), _p(o))
    return o


def f_invert(fd, a):
    a = _arr(a)
    o = np.zeros(4, dtype=np.uint64)
    ok = lib().kgo_f_invert(fd, _p(a), _p(o))
    return o if ok else None


def f_pow(fd, a, e):
    a, e = _arr(a), _arr(e)
    o = np.empty(4, dtype=np.uint64)
    lib().kgo_f_pow(fd, _p(a), _p(e), _p(o))
    return o


def f_consts(fd):
    o = np.empty(17, dtype=np.uint64)
    lib().kgo_f_consts(fd, _p(o))
    return {"p": o[0:4].copy(), "inv": int(o[4]), "r": o[5:9].copy(), "r2": o[9:13].copy(), "r3": o[13:17].copy()}


def f_vec_mul(fd, a, b):
    a, b = _arr(a), _arr(b)
    o = np.empty_like(a)
    lib().kgo_f_vec_mul(fd, _p(a), _p(b), _p(o), C.c_size_t(a.size // 4))
    return o


def f2_mul(a, b):
    a, b = _arr(a), _arr(b)
    o = np.empty(8, dtype=np.uint64)
    lib().kgo_f2_mul(_p(a), _p(b), _p(o))
    return o


def f2_square(a):
    a = _arr(a)
    o = np.empty(8, dtype=np.uint64)
    lib().kgo_f2_square(_p(a), _p(o))
    return o


def f2_invert(a):
    a = _arr(a)
    o = np.zeros(8, dtype=np.uint64)
    return o if lib().kgo_f2_invert(_p(a), _p(o)) else None


# ---- curves: "g1" (bn254 G1), "gk" (Grumpkin), "g2" (bn254 G2) -----------------------------------
NB = {"g1": 4, "gk": 4, "g2": 8}


def _fn(cv, name):
    return getattr(lib(), f"{cv}_kgo_{name}")


def generator(cv):
    o = np.empty(2 * NB[cv], dtype=np.uint64)
    _fn(cv, "generator")(_p(o))
    return o


def curve_b(cv):
    b, b3 = np.empty(NB[cv], dtype=np.uint64), np.empty(NB[cv], dtype=np.uint64)
    _fn(cv, "curve_b")(_p(b), _p(b3))
    return b, b3


def is_on_curve(cv, xy, inf=0):
    xy = _arr(xy)
    return bool(_fn(cv, "is_on_curve")(_p(xy), int(inf)))


def add_affine(cv, a, ainf, b, binf):
    a, b = _arr(a), _arr(b)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "add_affine")(_p(a), int(ainf), _p(b), int(binf), _p(o))
    return o


def double_affine(cv, a):
    a = _arr(a)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "double_affine")(_p(a), _p(o))
    return o


def add_mixed(cv, a, ainf, p):
    a, p = _arr(a), _arr(p)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "add_mixed")(_p(a), int(ainf), _p(p), _p(o))
    return o


def add_projective(cv, p, q):
    p, q = _arr(p), _arr(q)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "add_projective")(_p(p), _p(q), _p(o))
    return o


def double_projective(cv, p):
    p = _arr(p)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "double_projective")(_p(p), _p(o))
    return o


def scalar_point(cv, p, k_mont):
    p, k = _arr(p), _arr(k_mont)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "scalar_point")(_p(p), _p(k), _p(o))
    return o


def to_affine(cv, p):
    p = _arr(p)
    xy = np.empty(2 * NB[cv], dtype=np.uint64)
    inf = np.zeros(1, dtype=np.uint8)
    _fn(cv, "to_affine")(_p(p), _p(xy), _u8(inf))
    return xy, int(inf[0])


def msm(cv, bases, scalars, inf=None, threads=1):
    """groth16::msm_curve_addition restated; returns the PROJECTIVE result (3*NB limbs), n = min(len)."""
    bases, scalars = _arr(bases), _arr(scalars)
    n = min(bases.size // (2 * NB[cv]), scalars.size // 4)
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    o = np.empty(3 * NB[cv], dtype=np.uint64)
    _fn(cv, "msm")(_p(bases), _u8(inf), _p(scalars), C.c_size_t(n), _p(o), int(threads))
    return o


def commit_naive(cv, bases, scalars, inf=None):
    """nova PedersenCommitment::commit restated (naive NAF scalar muls); returns (xy, inf)."""
    bases, scalars = _arr(bases), _arr(scalars)
    n = min(bases.size // (2 * NB[cv]), scalars.size // 4)
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    xy = np.empty(2 * NB[cv], dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    _fn(cv, "commit_naive")(_p(bases), _u8(inf), _p(scalars), C.c_size_t(n), _p(xy), _u8(oinf))
    return xy, int(oinf[0])


# ---- Fft<Fr> ------------------------------------------------------------------------------------
class Fft:
    """groth16::fft::Fft<Fr> restated; operates on (n, 4) uint64 arrays, returns new arrays."""

    def __init__(self, k: int):
        self.k, self.n = k, 1 << k
        self._h = C.c_void_p(lib().kgo_fft_new(k))
        if not self._h:
            raise ValueError("k out of range")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kgo_fft_free(self._h)
            self._h = None

    def _pad(self, v):
        v = _arr(v).reshape(-1, 4)
        out = np.zeros((self.n, 4), dtype=np.uint64)
        out[: min(len(v), self.n)] = v[: self.n]
        return out

    def _run(self, name, v, threads):
        d = self._pad(v)
        getattr(lib(), name)(self._h, _p(d), int(threads))
        return d

    def dft(self, v, threads=1):
        return self._run("kgo_fft_dft", v, threads)

    def idft(self, v, threads=1):
        return self._run("kgo_fft_idft", v, threads)

    def coset_dft(self, v, threads=1):
        return self._run("kgo_fft_coset_dft", v, threads)

    def coset_idft(self, v, threads=1):
        return self._run("kgo_fft_coset_idft", v, threads)

    def divide_by_z_on_coset(self, v):
        d = self._pad(v)
        lib().kgo_fft_divide_by_z_on_coset(self._h, _p(d))
        return d


def fr_vec(name, a, b):
    a, b = _arr(a), _arr(b)
    o = np.empty_like(a)
    getattr(lib(), f"kgo_fr_vec_{name}")(_p(a), _p(b), _p(o), C.c_size_t(a.size // 4))
    return o


# ---- synthetic inputs ---------------------------------------------------------------------------
def gen_scalars(fd, seed, start, n):
    o = np.empty((n, 4), dtype=np.uint64)
    lib().kgo_gen_scalars(fd, C.c_uint64(seed), C.c_size_t(start), C.c_size_t(n), _p(o))
    return o


def gen_bases(curve, seed, start, n, threads=8):
    """curve: 0 = bn254 G1, 1 = Grumpkin.  Returns (n, 8) Montgomery x|y."""
    o = np.empty((n, 8), dtype=np.uint64)
    lib().kgo_gen_bases_mt(curve, C.c_uint64(seed), C.c_size_t(start), C.c_size_t(n), _p(o), int(threads))
    return o


# ---- Groth16 (oracle/kg_oracle_groth16.inc) -------------------------------------------------------
class R1cs:
    """Minimal R1CS container: CSR matrices over z = x || w (instance wires first), zkstd/src/r1cs.rs:11-27."""

    def __init__(self, a, b, c, x, w):
        self.a, self.b, self.c = a, b, c            # each: (row_ptr uint64[m+1], col uint64[nnz], val uint64[nnz,4])
        self.x, self.w = _arr(x).reshape(-1, 4), _arr(w).reshape(-1, 4)
        self.m, self.l, self.m_l_1 = len(a[0]) - 1, len(self.x), len(self.w)

    def evaluate(self):
        """cs.evaluate() (r1cs.rs:137-142): (A z, B z, C z)."""
        z = np.ascontiguousarray(np.concatenate([self.x, self.w]))
        outs = []
        for rp, col, val in (self.a, self.b, self.c):
            o = np.empty((self.m, 4), dtype=np.uint64)
            lib().kgo_r1cs_evaluate(_p(rp), _p(col), _p(val), C.c_size_t(self.m), _p(z), _p(o))
            outs.append(o)
        return outs


def chain_r1cs(m: int, t0_mont):
    """Synthetic circuit of SURVEY.md 8d(4): t_{i+1} = t_i * (t_i + 1), i < m; x = [1, t_0], w = [t_1 .. t_m].
    Constraint i: A = t_i, B = t_i + ONE, C = t_{i+1} (a FieldAssignment::mul gate on linear combinations)."""
    one = f_consts(FR)["r"]
    t = [_arr(t0_mont)]
    for _ in range(m):
        t.append(f_mul(FR, t[-1], f_add(FR, t[-1], one)))
    x = np.stack([one, t[0]])
    w = np.stack(t[1:])
    wire = lambda i: np.uint64(1) if i == 0 else np.uint64(2 + i - 1)        # t_0 is instance wire 1; t_i (i>0) is witness i-1 -> column l + (i-1), l = 2
    a_rp = np.arange(m + 1, dtype=np.uint64)
    a_col = np.array([wire(i) for i in range(m)], dtype=np.uint64)
    a_val = np.tile(one, (m, 1))
    b_rp = np.arange(0, 2 * m + 1, 2, dtype=np.uint64)
    b_col = np.empty(2 * m, dtype=np.uint64)
    b_col[0::2] = a_col
    b_col[1::2] = 0
    b_val = np.tile(one, (2 * m, 1))
    c_rp = np.arange(m + 1, dtype=np.uint64)
    c_col = np.array([wire(i + 1) for i in range(m)], dtype=np.uint64)
    c_val = np.tile(one, (m, 1))
    return R1cs((a_rp, a_col, np.ascontiguousarray(a_val)), (b_rp, b_col, np.ascontiguousarray(b_val)),
                (c_rp, c_col, np.ascontiguousarray(c_val)), x, w)


def groth16_setup_scalars(cs: R1cs, toxic):
    """zksnark.rs:17-127, scalar half.  toxic = (alpha, beta, gamma, delta, tau) Montgomery Fr, shape (5, 4)."""
    toxic = _arr(toxic).reshape(5, 4)
    nv = cs.l + cs.m_l_1
    h = np.zeros((max(cs.m - 1, 0), 4), dtype=np.uint64)
    lq = np.zeros((cs.m_l_1, 4), dtype=np.uint64)
    a = np.zeros((nv, 4), dtype=np.uint64)
    b = np.zeros((nv, 4), dtype=np.uint64)
    ic = np.zeros((cs.l, 4), dtype=np.uint64)
    rc = lib().kgo_groth16_setup_scalars(_p(cs.a[0]), _p(cs.a[1]), _p(cs.a[2]), _p(cs.b[0]), _p(cs.b[1]), _p(cs.b[2]),
                                         _p(cs.c[0]), _p(cs.c[1]), _p(cs.c[2]), C.c_size_t(cs.m), C.c_size_t(cs.l), C.c_size_t(cs.m_l_1),
                                         _p(toxic), _p(h), _p(lq), _p(a), _p(b), _p(ic))
    if rc:
        raise ValueError("ProverInversionFailed")
    return {"h": h, "l": lq, "a": a, "b": b, "ic": ic}


def fixed_base_mul(curve: int, k, threads=8):
    """generator * k -> (affine x|y, inf flags); curve 0 G1, 1 Grumpkin, 2 G2."""
    k = _arr(k).reshape(-1, 4)
    w = 16 if curve == 2 else 8
    xy = np.zeros((len(k), w), dtype=np.uint64)
    inf = np.zeros(len(k), dtype=np.uint8)
    if len(k):
        lib().kgo_fixed_base_mul(curve, _p(k), C.c_size_t(len(k)), _p(xy), _u8(inf), int(threads))
    return xy, inf


def groth16_params(cs: R1cs, toxic, threads=8):
    """Full Parameters (groth16/src/params.rs:6-28) by the oracle: scalars, then generator multiples."""
    sc = groth16_setup_scalars(cs, toxic)
    toxic = _arr(toxic).reshape(5, 4)
    P = {}
    P["h"], P["h_inf"] = fixed_base_mul(0, sc["h"], threads)
    P["l"], P["l_inf"] = fixed_base_mul(0, sc["l"], threads)
    P["a"], P["a_inf"] = fixed_base_mul(0, sc["a"], threads)
    P["b_g1"], P["b_g1_inf"] = fixed_base_mul(0, sc["b"], threads)
    P["b_g2"], P["b_g2_inf"] = fixed_base_mul(2, sc["b"], threads)
    P["ic"], P["ic_inf"] = fixed_base_mul(0, sc["ic"], threads)
    g1, _ = fixed_base_mul(0, np.stack([toxic[0], toxic[1], toxic[3]]), 1)        # alpha_g1, beta_g1, delta_g1
    g2, _ = fixed_base_mul(2, np.stack([toxic[1], toxic[3], toxic[2]]), 1)        # beta_g2, delta_g2, gamma_g2
    P["vk_g1"], P["vk_g2"] = g1, g2
    P["scalars"] = sc
    return P


def groth16_prove(cs: R1cs, P, r, s, threads=8, evals=None):
    """Prover::create_proof (prover.rs:20-99) with injected (r, s).  Returns (A xy[8], B xy[16], C xy[8], inf[3])."""
    a_ev, b_ev, c_ev = evals if evals is not None else cs.evaluate()
    out = np.zeros(32, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    r, s = _arr(r), _arr(s)
    rc = lib().kgo_groth16_prove(_p(a_ev), _p(b_ev), _p(c_ev), C.c_size_t(cs.m), _p(cs.x), C.c_size_t(cs.l), _p(cs.w), C.c_size_t(cs.m_l_1),
                                 _p(P["h"]), _u8(P["h_inf"]), _p(P["l"]), _u8(P["l_inf"]), _p(P["a"]), _u8(P["a_inf"]),
                                 _p(P["b_g1"]), _u8(P["b_g1_inf"]), _p(P["b_g2"]), _u8(P["b_g2_inf"]),
                                 _p(np.ascontiguousarray(P["vk_g1"])), _p(np.ascontiguousarray(P["vk_g2"][:2])), 0,
                                 _p(r), _p(s), _p(out), _u8(inf), int(threads))
    if rc:
        raise ValueError("ProverSubVersionCrsAttack")
    return out[:8].copy(), out[8:24].copy(), out[24:].copy(), inf


# ---- Nova (oracle/kg_oracle_nova.inc) ----------------------------------------------------------------
def matrix_prod(fd, csr, z):
    """SparseMatrix::prod (zkstd/src/matrix.rs:36-48) over z = (u | x | w)"""
    rp, col, val = (_arr(t) for t in csr)
    z = _arr(z)
    m = len(rp) - 1
    o = np.empty((m, 4), dtype=np.uint64)
    lib().kgo_matrix_prod(int(fd), _p(rp), _p(col), _p(val), C.c_size_t(m), _p(z), _p(o))
    return o


def nova_cross_term(fd, a, b, c, z1, z2, u1, u2):
    """Prover::compute_cross_term (nova/src/prover.rs:53-90): T = AZ1 o BZ2 + AZ2 o BZ1 - u1 CZ2 - u2 CZ1"""
    a, b, c = ([_arr(t) for t in csr] for csr in (a, b, c))
    z1, z2, u1, u2 = _arr(z1), _arr(z2), _arr(u1), _arr(u2)
    m = len(a[0]) - 1
    o = np.empty((m, 4), dtype=np.uint64)
    lib().kgo_nova_cross_term(int(fd), _p(a[0]), _p(a[1]), _p(a[2]), _p(b[0]), _p(b[1]), _p(b[2]), _p(c[0]), _p(c[1]), _p(c[2]),
                              C.c_size_t(m), _p(z1), _p(z2), _p(u1), _p(u2), _p(o))
    return o

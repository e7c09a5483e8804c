"""Deterministic synthetic inputs of SURVEY.md 8d(4), shared by bench.py and the full-size parity tests: the chain
circuit t_{i+1} = t_i * (t_i + 1) as an R1CS of m multiplication constraints (FieldAssignment::mul semantics,
zkstd/src/circuit/gadget/field.rs:48-61), its witness, a fixed toxic waste and fixed blinding scalars.

Plain numpy / Python integers only (inputs, not arithmetic of the path): field elements are 4 little-endian uint64
Montgomery limbs, matrices are CSR triples over z = x || w (instance wires [1, t_0] first)."""
from __future__ import annotations

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_MASK = 0xFFFFFFFFFFFFFFFF


def mont(v: int) -> list[int]:
    """canonical integer -> Montgomery limbs (R = 2^256)"""
    v = (v << 256) % R_MOD
    return [(v >> (64 * j)) & _MASK for j in range(4)]


def mont_vec(vals) -> np.ndarray:
    return np.array([mont(v) for v in vals], dtype=np.uint64).reshape(-1, 4)


class FrOps:
    """the five host-side scalar operations api.groth16_setup asks for, in plain Python integers"""
    _rinv = pow(1 << 256, -1, R_MOD)
    to_i = staticmethod(lambda v: (sum(int(x) << (64 * j) for j, x in enumerate(v)) * FrOps._rinv) % R_MOD)
    to_m = staticmethod(lambda i: np.array(mont(i % R_MOD), dtype=np.uint64))
    one = staticmethod(lambda: np.array(mont(1), dtype=np.uint64))
    inv = classmethod(lambda cls, x: cls.to_m(pow(cls.to_i(x), -1, R_MOD)))
    mul = classmethod(lambda cls, x, y: cls.to_m(cls.to_i(x) * cls.to_i(y)))
    sub = classmethod(lambda cls, x, y: cls.to_m(cls.to_i(x) - cls.to_i(y)))
    pow2k = classmethod(lambda cls, x, k: cls.to_m(pow(cls.to_i(x), 1 << k, R_MOD)))


class ChainCircuit:
    """m constraints  A_i = t_i, B_i = t_i + 1, C_i = t_{i+1};  x = [1, t_0] (l = 2), w = [t_1 .. t_m] (m_l_1 = m)."""

    def __init__(self, m: int, t0: int = 0x123456789ABCDEF0123456789ABCDEF):
        self.m, self.l, self.m_l_1 = m, 2, m
        t = [t0 % R_MOD]
        for _ in range(m):
            t.append(t[-1] * (t[-1] + 1) % R_MOD)
        tm = mont_vec(t)
        one = np.array(mont(1), dtype=np.uint64)
        self.a_eval = tm[:m].copy()                                     # cs.evaluate() (zkstd/src/r1cs.rs:137-142)
        self.b_eval = mont_vec([(v + 1) % R_MOD for v in t[:m]])
        self.c_eval = tm[1:].copy()
        self.x = np.stack([one, tm[0]])
        self.w = tm[1:].copy()
        wire = np.concatenate([[1], 2 + np.arange(m, dtype=np.uint64)]).astype(np.uint64)   # t_0: instance wire 1; t_i: witness i-1
        ones = np.tile(one, (m, 1))
        self.a = (np.arange(m + 1, dtype=np.uint64), wire[:m].copy(), ones)
        b_col = np.empty(2 * m, dtype=np.uint64)
        b_col[0::2] = wire[:m]
        b_col[1::2] = 0
        self.b = (np.arange(0, 2 * m + 1, 2, dtype=np.uint64), b_col, np.tile(one, (2 * m, 1)))
        self.c = (np.arange(m + 1, dtype=np.uint64), wire[1:].copy(), ones)


def witness_like(scal: np.ndarray, seed: int) -> np.ndarray:
    """Scalars distributed like a Groth16 aux vector (groth16/src/prover.rs:53-65 meets bit decompositions and selectors): half
    ones, a fifth zeros, a tenth -1, a tenth the small value 5, a tenth left as they are (uniform).  In place on an (n, 4) array
    of Montgomery limbs; the same distribution as tests/test_gpu_large.py::_witness_like."""
    n = len(scal)
    kind = np.random.default_rng(seed).integers(0, 10, n)
    scal[kind < 5] = np.array(mont(1), dtype=np.uint64)
    scal[(kind >= 5) & (kind < 7)] = 0
    scal[kind == 7] = np.array(mont(R_MOD - 1), dtype=np.uint64)
    scal[kind == 8] = np.array(mont(5), dtype=np.uint64)
    return scal


class BooleanHeavyCircuit:
    """A circuit whose witness looks like a real one: mc chain constraints t_{i+1} = t_i (t_i + 1) (uniform wires) followed by
    mb booleanity constraints b (b - 1) = 0 on wires that are 1 (5 of 7) or 0 (2 of 7) -- z = x || w is then half ones, a fifth
    zeros, three tenths uniform.  Same interface as ChainCircuit; x = [1, t_0], w = [t_1 .. t_mc, b_1 .. b_mb]."""

    def __init__(self, m: int, t0: int = 0x123456789ABCDEF0123456789ABCDEF, seed: int = 7):
        mc = (3 * m) // 10
        mb = m - mc
        self.m, self.l, self.m_l_1 = m, 2, m
        t = [t0 % R_MOD]
        for _ in range(mc):
            t.append(t[-1] * (t[-1] + 1) % R_MOD)
        bits = (np.random.default_rng(seed).integers(0, 7, mb) < 5).astype(np.int64)
        one = np.array(mont(1), dtype=np.uint64)
        zero = np.zeros(4, dtype=np.uint64)
        minus_one = np.array(mont(R_MOD - 1), dtype=np.uint64)
        tm = mont_vec(t)
        bm = np.where(bits[:, None] == 1, one[None, :], zero[None, :]).astype(np.uint64)
        self.a_eval = np.concatenate([tm[:mc], bm])                                          # A z: t_i | b_j
        self.b_eval = np.concatenate([mont_vec([(v + 1) % R_MOD for v in t[:mc]]),           # B z: t_i + 1 | b_j - 1
                                      np.where(bits[:, None] == 1, zero[None, :], minus_one[None, :]).astype(np.uint64)])
        self.c_eval = np.concatenate([tm[1:], np.zeros((mb, 4), dtype=np.uint64)])           # C z: t_{i+1} | 0
        self.x = np.stack([one, tm[0]])
        self.w = np.concatenate([tm[1:], bm])
        wire_t = np.concatenate([[1], 2 + np.arange(mc, dtype=np.uint64)]).astype(np.uint64)    # t_0: instance wire 1; t_i: witness i-1
        wire_b = (2 + mc + np.arange(mb, dtype=np.uint64)).astype(np.uint64)
        self.a = (np.arange(m + 1, dtype=np.uint64), np.concatenate([wire_t[:mc], wire_b]), np.tile(one, (m, 1)))
        b_col = np.empty(2 * m, dtype=np.uint64)
        b_col[0::2] = np.concatenate([wire_t[:mc], wire_b])
        b_col[1::2] = 0
        b_val = np.tile(one, (2 * m, 1))
        b_val[2 * mc + 1::2] = minus_one                                                     # (b_j) + (-1) * 1
        self.b = (np.arange(0, 2 * m + 1, 2, dtype=np.uint64), b_col, b_val)
        c_ptr = np.concatenate([np.arange(mc + 1, dtype=np.uint64), np.full(mb, mc, dtype=np.uint64)])
        self.c = (c_ptr, wire_t[1:].copy(), np.tile(one, (mc, 1)))


def fixed_toxic() -> np.ndarray:
    """alpha, beta, gamma, delta, tau (the reference draws them from its rng, groth16/src/zksnark.rs:28-32)"""
    return mont_vec([(0xA11CE + 0x9E3779B97F4A7C15 * (j + 1)) ** 3 % R_MOD for j in range(5)])


def fixed_rs():
    """(r, s): the prover's blinding scalars (prover.rs:71-72)"""
    r = np.array(mont(0x1111111111111111222222222222222233333333333333334444444444444 % R_MOD), dtype=np.uint64)
    s = np.array(mont(0x5555555555555555666666666666666677777777777777778888888888888 % R_MOD), dtype=np.uint64)
    return r, s

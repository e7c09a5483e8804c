"""Shared helpers for the parity tests (numpy <-> big-int conversions)."""
import ctypes as C

import numpy as np

from oracle import pyoracle as P

U32P = C.POINTER(C.c_uint32)
U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)


def p32(a):
    return a.ctypes.data_as(U32P)


def L(x):
    return np.array(P.limbs(x), dtype=np.uint64)


def I(a):
    return P.from_limbs([int(v) for v in a])


def pt_to_np(cur, pt):
    """affine big-int point -> reference-form limbs (x | y), zeros for the identity."""
    p = cur.p
    if pt is None:
        return np.zeros(16 if cur.ext else 8, dtype=np.uint64)
    if cur.ext:
        x, y = pt
        return np.concatenate([L(P.to_mont(v, p)) for v in (x.a, x.b, y.a, y.b)])
    return np.concatenate([L(P.to_mont(pt[0], p)), L(P.to_mont(pt[1], p))])


def np_to_pt(cur, xy, inf):
    p = cur.p
    if inf:
        return None
    if cur.ext:
        v = [P.from_mont(I(xy[4 * i:4 * i + 4]), p) for i in range(4)]
        return (P.Fq2(v[0], v[1]), P.Fq2(v[2], v[3]))
    return (P.from_mont(I(xy[:4]), p), P.from_mont(I(xy[4:]), p))


CURVES = {"g1": (0, P.G1, 0), "gk": (1, P.GRUMPKIN, 1), "g2": (2, P.G2, 0)}  # name -> (id, pycurve, scalar field id)


def digest_limbs(arr, p):
    """sha256 over the canonical integers (32-byte little endian each) of an (n, 4) array of Montgomery limbs -- the
    digest format of tests/golden/big.json (make_golden.py: digest)."""
    import hashlib
    h = hashlib.sha256()
    rinv = pow(1 << 256, -1, p)
    for row in np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4):
        h.update((I(row) * rinv % p).to_bytes(32, "little"))
    return h.hexdigest()


class PyFrOps:
    """the five host-side Fr scalar operations kogarashi_amd.api.groth16_setup asks for, in plain Python integers"""
    p = P.R_MOD

    @classmethod
    def _i(cls, v):
        return P.from_mont(I(v), cls.p)

    @classmethod
    def _m(cls, i):
        return L(P.to_mont(i % cls.p, cls.p))

    @classmethod
    def one(cls):
        return cls._m(1)

    @classmethod
    def inv(cls, x):
        return cls._m(pow(cls._i(x), -1, cls.p))

    @classmethod
    def mul(cls, x, y):
        return cls._m(cls._i(x) * cls._i(y))

    @classmethod
    def sub(cls, x, y):
        return cls._m(cls._i(x) - cls._i(y))

    @classmethod
    def pow2k(cls, x, k):
        return cls._m(pow(cls._i(x), 1 << k, cls.p))

"""Independent big-integer oracle for the BN254 / Grumpkin MSM + NTT + Groth16 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as
the checker.

This file deliberately does NOT follow the reference's limb code: it computes every result with plain
Python integers (``pow(x, -1, p)``, affine chord-and-tangent), so that the C restatement
(``oracle/kg_oracle.c``, which *does* follow the reference line by line) and the HIP kernels are both
checked against an implementation that shares no code and no algorithm with them.

PARITY PINNING.  The reference (KogarashiNetwork/Kogarashi, Rust) cannot be built here (no
rustc/cargo, un-vendored crates) and holds NO known-answer vectors for Montgomery multiplication,
point addition, MSM, NTT or Groth16.  What it does hold are constants, and those are
pinned in ``tests/test_oracle_constants.py``:
  bn254/src/fr.rs:10-65 (MODULUS, R, R2, R3, INV, S, ROOT_OF_UNITY, GENERATOR=7),
  bn254/src/fq.rs:9-42 (MODULUS, R, R2, R3, INV), bn254/src/params.rs:8-57 (G1/G2 generators, b),
  grumpkin/src/params.rs:4-19 (generator, b = -17).
Every output on the path is a canonical value (reduced Montgomery limbs / affine points), so
"bit-exact" is defined mathematically and this oracle defines it.
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------
# constants (values as integers; the hex strings are the ones quoted in the reference)
# --------------------------------------------------------------------------------------
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr, bn254/src/fr.rs:10
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq, bn254/src/fq.rs:9
MONT_R = 1 << 256
FR_S = 28                                   # two-adicity, bn254/src/fr.rs:53
FR_GENERATOR = 7                            # bn254/src/fr.rs:18
# ROOT_OF_UNITY (canonical integer; the reference stores to_mont_form of it), bn254/src/fr.rs:60-65
FR_ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C

MASK64 = (1 << 64) - 1


def to_mont(x: int, p: int) -> int:
    return (x * MONT_R) % p


def from_mont(x: int, p: int) -> int:
    return (x * pow(MONT_R, -1, p)) % p


def limbs(x: int, n: int = 4) -> list[int]:
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def from_limbs(l) -> int:
    v = 0
    for i, w in enumerate(l):
        v |= int(w) << (64 * i)
    return v


def mont_inv64(p: int) -> int:
    """INV = -p^{-1} mod 2^64."""
    return (-pow(p, -1, 1 << 64)) % (1 << 64)


# --------------------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2+1)   (bn254/src/fqn.rs:12-13)
# --------------------------------------------------------------------------------------
class Fq2:
    __slots__ = ("a", "b")

    def __init__(self, a: int, b: int = 0):
        self.a = a % Q_MOD
        self.b = b % Q_MOD

    def __add__(self, o):
        return Fq2(self.a + o.a, self.b + o.b)

    def __sub__(self, o):
        return Fq2(self.a - o.a, self.b - o.b)

    def __neg__(self):
        return Fq2(-self.a, -self.b)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fq2(self.a * o, self.b * o)
        return Fq2(self.a * o.a - self.b * o.b, self.a * o.b + self.b * o.a)

    def __eq__(self, o):
        return self.a == o.a and self.b == o.b

    def __hash__(self):
        return hash((self.a, self.b))

    def is_zero(self):
        return self.a == 0 and self.b == 0

    def inv(self):
        t = pow(self.a * self.a + self.b * self.b, -1, Q_MOD)
        return Fq2(self.a * t, -self.b * t)

    def __repr__(self):
        return f"Fq2({hex(self.a)}, {hex(self.b)})"


# --------------------------------------------------------------------------------------
# curves: y^2 = x^3 + b, affine with None = point at infinity
# --------------------------------------------------------------------------------------
class Curve:
    """Short-Weierstrass a = 0 curve over a prime field (ints mod p) or over Fq2."""

    def __init__(self, name, p_base, p_scalar, b, gen, ext=False):
        self.name, self.p, self.n, self.b, self.gen, self.ext = name, p_base, p_scalar, b, gen, ext

    # base-field helpers -------------------------------------------------------------
    def _inv(self, x):
        return x.inv() if self.ext else pow(x, -1, self.p)

    def _red(self, x):
        return x if self.ext else x % self.p

    def _is_zero(self, x):
        return x.is_zero() if self.ext else x % self.p == 0

    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return self._is_zero(y * y - (x * x * x + self.b))

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self._red(-P[1]))

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if self._is_zero(x1 - x2):
            if self._is_zero(y1 - y2):
                if self._is_zero(y1):
                    return None
                lam = (x1 * x1 * 3) * self._inv(y1 * 2)
            else:
                return None
        else:
            lam = (y2 - y1) * self._inv(x2 - x1)
        x3 = self._red(lam * lam - x1 - x2)
        y3 = self._red(lam * (x1 - x3) - y1)
        return (x3, y3)

    def mul(self, P, k: int):
        k %= self.n
        acc = None
        add = P
        while k:
            if k & 1:
                acc = self.add(acc, add)
            add = self.add(add, add)
            k >>= 1
        return acc

    def msm(self, points, scalars):
        """sum_i k_i * P_i over min(len) pairs (groth16/src/msm.rs:25 zip semantics), bucketed for speed."""
        n = min(len(points), len(scalars))
        if n == 0:
            return None
        c = 8 if n >= 64 else 4
        nwin = (256 + c - 1) // c
        acc = None
        for w in reversed(range(nwin)):
            for _ in range(c):
                acc = self.add(acc, acc)
            buckets = [None] * (1 << c)
            for i in range(n):
                d = ((scalars[i] % self.n) >> (w * c)) & ((1 << c) - 1)
                if d:
                    buckets[d] = self.add(buckets[d], points[i])
            run = None
            tot = None
            for d in range((1 << c) - 1, 0, -1):
                run = self.add(run, buckets[d])
                tot = self.add(tot, run)
            acc = self.add(acc, tot)
        return acc

    def msm_naive(self, points, scalars):
        acc = None
        for P, k in zip(points, scalars):
            acc = self.add(acc, self.mul(P, k))
        return acc


G1 = Curve("bn254_g1", Q_MOD, R_MOD, 3, (1, 2))                      # bn254/src/params.rs:8-12
# Grumpkin: y^2 = x^3 - 17 over Fr, scalars in Fq (grumpkin/src/params.rs:4-19)
_GRUMPKIN_GEN_Y_MONT = 0x14B34CF69DC25D68AA7B8CF435DFAFBB23D3446F21C77DC311B2DFF1448C41D8
GRUMPKIN = Curve("grumpkin", R_MOD, Q_MOD, R_MOD - 17, (1, from_mont(_GRUMPKIN_GEN_Y_MONT, R_MOD)))
# G2: y^2 = x^3 + 3/(9+u) over Fq2 (bn254/src/params.rs:15-57)
G2_B = Fq2(0x2B149D40CEB8AAAE81BE18991BE06AC3B5B4C5E559DBEFA33267E6DC24A138E5,
           0x009713B03AF0FED4CD2CAFADEED8FDF4A74FA084E52D1852E4A2BD0685C315D2)
G2_GEN = (Fq2(0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
              0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
          Fq2(0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
              0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B))
G2 = Curve("bn254_g2", Q_MOD, R_MOD, G2_B, G2_GEN, ext=True)


# --------------------------------------------------------------------------------------
# NTT over Fr, natural order in / natural order out  (groth16/src/fft.rs:92-127)
# --------------------------------------------------------------------------------------
def fr_omega(k: int) -> int:
    """n-th root of unity the reference uses for n = 2^k: ROOT_OF_UNITY^(2^(S-k)) (fft.rs:34)."""
    assert 1 <= k <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD)


def _ntt(v, w, p):
    n = len(v)
    if n == 1:
        return list(v)
    even = _ntt(v[0::2], w * w % p, p)
    odd = _ntt(v[1::2], w * w % p, p)
    out = [0] * n
    t = 1
    h = n // 2
    for i in range(h):
        x = t * odd[i] % p
        out[i] = (even[i] + x) % p
        out[i + h] = (even[i] - x) % p
        t = t * w % p
    return out


def ntt_naive(v, k):
    n = 1 << k
    w = fr_omega(k)
    v = list(v) + [0] * (n - len(v))
    return [sum(v[j] * pow(w, i * j, R_MOD) for j in range(n)) % R_MOD for i in range(n)]


def dft(v, k):
    n = 1 << k
    v = [x % R_MOD for x in v] + [0] * (n - len(v))
    return _ntt(v, fr_omega(k), R_MOD)


def idft(v, k):
    n = 1 << k
    v = [x % R_MOD for x in v] + [0] * (n - len(v))
    ninv = pow(n, -1, R_MOD)
    return [x * ninv % R_MOD for x in _ntt(v, pow(fr_omega(k), -1, R_MOD), R_MOD)]


def coset_dft(v, k):
    g = 1
    out = []
    for x in v:
        out.append(x * g % R_MOD)
        g = g * FR_GENERATOR % R_MOD
    return dft(out, k)


def coset_idft(v, k):
    ginv = pow(FR_GENERATOR, -1, R_MOD)
    g = 1
    out = []
    for x in idft(v, k):
        out.append(x * g % R_MOD)
        g = g * ginv % R_MOD
    return out


def divide_by_z_on_coset(v, k):
    zi = pow(pow(FR_GENERATOR, 1 << k, R_MOD) - 1, -1, R_MOD)
    return [x * zi % R_MOD for x in v]


# --------------------------------------------------------------------------------------
# deterministic synthetic inputs: splitmix64 stream, wide reduction, try-and-increment
# --------------------------------------------------------------------------------------
SEED_BASE = 0x4B6F676172617368  # "Kogarash"


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)


def stream_at(seed: int, index: int) -> SplitMix64:
    """Element `index` of stream `seed` owns 8 consecutive splitmix64 outputs; random access by
    jumping the additive state, so the GPU generator and the CPU oracle produce identical values."""
    g = SplitMix64(0)
    g.s = (seed + 8 * index * 0x9E3779B97F4A7C15) & MASK64
    return g


def scalar_at(seed: int, index: int, p: int) -> int:
    """Uniform mod p by the reference's wide reduction of 8 u64 words
    (zkstd/src/arithmetic/limbs/bits_256/represent.rs:80-103): (lo + hi*2^256) mod p.  Canonical int."""
    g = stream_at(seed, index)
    w = [g.next() for _ in range(8)]
    return (from_limbs(w[:4]) + (from_limbs(w[4:]) << 256)) % p


def _sqrt_mod(a: int, p: int):
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    # Tonelli-Shanks
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, tt, r = s, pow(z, t, p), pow(a, t, p), pow(a, (t + 1) // 2, p)
    while tt != 1:
        i, x = 0, tt
        while x != 1:
            x = x * x % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        tt, r = tt * c % p, r * b % p
    return r


def base_at(curve: Curve, seed: int, index: int):
    """Valid curve point by try-and-increment on a seeded x (prime-field curves only).
    y is the root returned by a^((p+1)/4) for p = 3 mod 4 (bn254/src/fq.rs:121-127 exponent); for Grumpkin
    (p = 1 mod 4) the smaller of the two roots is taken so the choice is algorithm independent; the
    low bit of the 5th stream word then decides whether y is negated."""
    assert not curve.ext
    p = curve.p
    g = stream_at(seed, index)
    w = [g.next() for _ in range(8)]
    x = from_limbs(w[:4]) % p
    flip = w[4] & 1
    while True:
        y = _sqrt_mod(x * x * x + curve.b, p)
        if y is not None and y != 0:
            if p % 4 != 3:
                y = min(y, p - y)
            if flip:
                y = p - y
            return (x, y)
        x = (x + 1) % p

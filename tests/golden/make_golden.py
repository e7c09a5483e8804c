#!/usr/bin/env python3
"""Generates tests/golden/*.json from the independent big-integer oracle (oracle/pyoracle.py).

The reference is Rust and cannot be imported or built here, and holds no known-answer vectors for this path
(SURVEY.md 4/8c), so these vectors come from plain Python integer arithmetic that shares no code with either the
C restatement or the HIP kernels.  Values are canonical integers in hex; tests convert to Montgomery limbs.
Run from the repo root:  python tests/golden/make_golden.py"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as P  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = P.SEED_BASE


def hx(v):
    return hex(v)


def pt(cur, p):
    if p is None:
        return None
    if cur.ext:
        return [hx(p[0].a), hx(p[0].b), hx(p[1].a), hx(p[1].b)]
    return [hx(p[0]), hx(p[1])]


def field_vectors():
    rnd = random.Random(11)
    out = {}
    for name, p in (("fr", P.R_MOD), ("fq", P.Q_MOD)):
        rows = []
        specials = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, P.MONT_R % p, pow(P.MONT_R, -1, p)]
        pairs = [(a, b) for a in specials for b in specials[:4]] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(40)]
        for a, b in pairs:
            rows.append({"a": hx(a), "b": hx(b), "add": hx((a + b) % p), "sub": hx((a - b) % p), "mul": hx(a * b % p),
                         "neg": hx(-a % p), "dbl": hx(2 * a % p), "sqr": hx(a * a % p), "inv": hx(pow(a, -1, p)) if a else None})
        out[name] = rows
    return out


def curve_points(cur, k, seed):
    if cur.ext:
        rnd = random.Random(seed)
        return [cur.mul(cur.gen, rnd.randrange(1, cur.n)) for _ in range(k)]
    return [P.base_at(cur, seed, i) for i in range(k)]


def msm_vectors():
    out = {}
    for name, cur, seed in (("g1", P.G1, SEED + 1), ("gk", P.GRUMPKIN, SEED + 2), ("g2", P.G2, SEED + 3)):
        cases = []
        sizes = (1, 2, 3, 4, 31, 32, 33) if not cur.ext else (1, 2, 4, 9)
        for n in sizes:
            rnd = random.Random(seed + n)
            pts = curve_points(cur, n, seed + n)
            ks = [rnd.randrange(cur.n) for _ in range(n)]
            inf = [0] * n
            if n >= 4:
                inf[2] = 1
                ks[1] = 0
                pts[3] = pts[0]
                ks[0] = cur.n - 1
            live = [None if f else q for q, f in zip(pts, inf)]
            res = cur.msm_naive(live, ks)
            cases.append({"n": n, "bases": [pt(cur, q) for q in pts], "inf": inf, "scalars": [hx(k) for k in ks], "sum": pt(cur, res)})
        # length mismatch: zip semantics (msm.rs:25)
        pts = curve_points(cur, 5, seed + 99)
        ks = [3, 5, 7]
        cases.append({"n": 3, "bases": [pt(cur, q) for q in pts], "inf": [0] * 5, "scalars": [hx(k) for k in ks],
                      "sum": pt(cur, cur.msm_naive(pts[:3], ks))})
        out[name] = cases
    return out


def point_vectors():
    out = {}
    for name, cur, seed in (("g1", P.G1, SEED + 4), ("gk", P.GRUMPKIN, SEED + 5), ("g2", P.G2, SEED + 6)):
        a, b = curve_points(cur, 2, seed)
        rows = []
        for x, y in ((a, b), (a, a), (a, cur.neg(a)), (a, None), (None, b), (None, None), (cur.add(a, a), a)):
            rows.append({"p": pt(cur, x), "q": pt(cur, y), "sum": pt(cur, cur.add(x, y))})
        out[name] = {"add": rows, "gen": pt(cur, cur.gen), "gen_times_23": pt(cur, cur.mul(cur.gen, 23)),
                     "seven_plus_sixteen": pt(cur, cur.add(cur.mul(cur.gen, 7), cur.mul(cur.gen, 16)))}
    return out


def ntt_vectors():
    out = []
    for k in (1, 2, 3, 6):
        rnd = random.Random(100 + k)
        n = 1 << k
        v = [rnd.randrange(P.R_MOD) for _ in range(n)]
        if k >= 2:
            v[1] = 0
            v[-1] = P.R_MOD - 1
        assert P.dft(v, k) == P.ntt_naive(v, k)
        out.append({"k": k, "v": [hx(x) for x in v], "dft": [hx(x) for x in P.dft(v, k)], "idft": [hx(x) for x in P.idft(v, k)],
                    "coset_dft": [hx(x) for x in P.coset_dft(v, k)], "coset_idft": [hx(x) for x in P.coset_idft(v, k)],
                    "divide_by_z_on_coset": [hx(x) for x in P.divide_by_z_on_coset(v, k)]})
    return out


def digest(ints):
    """sha256 over the canonical integers as 32-byte little-endian strings (what the tests recompute from limbs)"""
    import hashlib
    h = hashlib.sha256()
    for v in ints:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def big_vectors():
    """MSM n = 2^10 (BASELINE.json configs[0]), NTT k = 10 and one tiny Groth16 proof with
    fixed (r, s).  Inputs of the two large cases are the seeded streams every implementation regenerates (base_at /
    scalar_at == kg_gen_bases / kg_gen_scalars == the oracle's generators); the fixture pins them with digests and holds
    the outputs (the MSM sum; digests plus sampled entries of the transforms)."""
    out = {}
    n = 1 << 10
    seed_b, seed_s = SEED + 0x1000, SEED + 0x1001
    pts = [P.base_at(P.G1, seed_b, i) for i in range(n)]
    ks = [P.scalar_at(seed_s, i, P.R_MOD) for i in range(n)]
    res = P.G1.msm(pts, ks)
    assert res == P.G1.msm_naive(pts, ks)                      # bucketed == sum of independent double-and-adds
    out["msm_2_10"] = {"curve": "g1", "n": n, "seed_bases": hx(seed_b), "seed_scalars": hx(seed_s),
                       "bases_digest": digest([c for q in pts for c in q]), "scalars_digest": digest(ks), "sum": pt(P.G1, res)}
    k = 10
    seed_v = SEED + 0x1002
    v = [P.scalar_at(seed_v, i, P.R_MOD) for i in range(1 << k)]
    ntt = {"k": k, "seed": hx(seed_v), "input_digest": digest(v)}
    sample = [0, 1, 2, 511, 512, 1022, 1023]
    for name, fn in (("dft", P.dft), ("idft", P.idft), ("coset_dft", P.coset_dft), ("coset_idft", P.coset_idft)):
        o = fn(v, k)
        ntt[name] = {"digest": digest(o), "sample_index": sample, "sample": [hx(o[i]) for i in sample]}
    assert P.idft(P.dft(v, k), k) == v
    out["ntt_2_10"] = ntt
    out["groth16_tiny"] = groth16_vector()
    return out


def groth16_vector():
    """One Groth16 proof with everything fixed, computed with integers only: with the toxic waste known the discrete
    logs of the CRS elements and of the proof follow from the witness, and every point is that multiple of its
    generator (groth16/src/zksnark.rs:17-127 setup, prover.rs:20-99 create_proof, read in the exponent).  Circuit: the
    4-constraint chain t_{i+1} = t_i (t_i + 1) (the size of the reference's own end-to-end test, groth16/src/lib.rs:29-77)."""
    p = P.R_MOD
    m, l = 4, 2
    t = [0x1234567 % p]
    for _ in range(m):
        t.append(t[-1] * (t[-1] + 1) % p)
    z = [1, t[0]] + t[1:]                                   # x = [1, t_0], w = t_1..t_m
    wire = lambda i: 1 if i == 0 else 2 + i - 1
    A = [{wire(i): 1} for i in range(m)]
    B = [{wire(i): 1, 0: 1} for i in range(m)]
    C = [{wire(i + 1): 1} for i in range(m)]
    ev = lambda M: [sum(c * z[v] for v, c in row.items()) % p for row in M]
    av, bv, cv = ev(A), ev(B), ev(C)
    assert all(x * y % p == w for x, y, w in zip(av, bv, cv))
    alpha, beta, gamma, delta, tau = [pow(0xA11CE + 977 * j, 5, p) for j in range(1, 6)]
    r, s = 0x1111111122222222333333334444444455555555 % p, 0x66666666777777778888888899999999AAAAAAAA % p
    k, n = 2, 4
    lag = P.idft([pow(tau, i, p) for i in range(m)] + [0] * (n - m), k)      # zksnark.rs:44-49,61
    nv = len(z)
    at_tau = lambda M: [sum(row.get(v, 0) * lag[i] for i, row in enumerate(M)) % p for v in range(nv)]
    a_s, b_s, c_s = at_tau(A), at_tau(B), at_tau(C)
    ext = [(beta * a + alpha * b + c) % p for a, b, c in zip(a_s, b_s, c_s)]
    ic_s = [e * pow(gamma, -1, p) % p for e in ext[:l]]
    l_s = [e * pow(delta, -1, p) % p for e in ext[l:]]
    coeff = (pow(tau, n, p) - 1) * pow(delta, -1, p) % p
    h_s = [pow(tau, i, p) * coeff % p for i in range(m - 1)]
    ac, bc, cc = (P.coset_dft(P.idft(v, k), k) for v in (av, bv, cv))
    q = P.coset_idft(P.divide_by_z_on_coset([(x * y - w) % p for x, y, w in zip(ac, bc, cc)], k), k)
    a_dl = (alpha + sum(zi * ai for zi, ai in zip(z, a_s)) + r * delta) % p
    b_dl = (beta + sum(zi * bi for zi, bi in zip(z, b_s)) + s * delta) % p
    c_dl = (sum(zi * li for zi, li in zip(z[l:], l_s)) + sum(qi * hi for qi, hi in zip(q, h_s)) + s * a_dl + r * b_dl - r * s * delta) % p
    # the Groth16 verification equation in the exponent (groth16/src/verifier.rs): e(A, B) = e(alpha, beta) e(ic . x, gamma) e(C, delta)
    assert a_dl * b_dl % p == (alpha * beta + sum(zi * ici for zi, ici in zip(z[:l], ic_s)) * gamma + c_dl * delta) % p
    g1m = lambda e: pt(P.G1, P.G1.mul(P.G1.gen, e))
    g2m = lambda e: pt(P.G2, P.G2.mul(P.G2.gen, e))
    return {"m": m, "l": l, "t0": hx(t[0]), "toxic": [hx(v) for v in (alpha, beta, gamma, delta, tau)], "r": hx(r), "s": hx(s),
            "crs_scalars": {"h": [hx(v) for v in h_s], "l": [hx(v) for v in l_s], "a": [hx(v) for v in a_s], "b": [hx(v) for v in b_s],
                            "ic": [hx(v) for v in ic_s]},
            "h_coefficients": [hx(v) for v in q],
            "proof": {"a": g1m(a_dl), "b": g2m(b_dl), "c": g1m(c_dl)}}


def main():
    data = {"field": field_vectors(), "points": point_vectors(), "msm": msm_vectors(), "ntt": ntt_vectors(), "big": big_vectors()}
    for k, v in data.items():
        with open(os.path.join(HERE, f"{k}.json"), "w") as f:
            json.dump(v, f, indent=0, separators=(",", ":"))
        print(k, os.path.getsize(os.path.join(HERE, f"{k}.json")), "bytes")


if __name__ == "__main__":
    main()

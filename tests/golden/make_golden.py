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


def main():
    data = {"field": field_vectors(), "points": point_vectors(), "msm": msm_vectors(), "ntt": ntt_vectors()}
    for k, v in data.items():
        with open(os.path.join(HERE, f"{k}.json"), "w") as f:
            json.dump(v, f, indent=0, separators=(",", ":"))
        print(k, os.path.getsize(os.path.join(HERE, f"{k}.json")), "bytes")


if __name__ == "__main__":
    main()

"""Randomised parity stress of the MSM on skewed scalar vectors (one-off check of the hot-bucket paths; needs the oracle):
   python tools/dbg/stress_skew.py [rounds]
Patterns: all ones; two values; small values (< 2^k); witness-like; a few huge buckets among uniform scalars; all equal random; zeros."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
from kogarashi_amd import synthetic as syn
from oracle import oracle as O
K.init()
ctx = K.Context(0)
rng = np.random.default_rng(int(os.environ.get("KG_STRESS_SEED", "7")))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
names = {0: "g1", 1: "gk", 2: "g2"}
bad = 0
for it in range(rounds):
    curve = int(rng.choice([0, 0, 0, 1, 2]))
    lo_lg, hi_lg = [int(v) for v in os.environ.get("KG_STRESS_LG", "12,19").split(",")]       # sizes 2^lo .. 2^hi - 1 (G2: two fewer)
    lg = int(rng.integers(lo_lg, hi_lg if curve != 2 else max(lo_lg + 1, hi_lg - 2)))
    n = max(1, (1 << lg) + int(rng.integers(-50, 50)))
    fld = 1 if curve == 1 else 0
    bases = O.gen_bases(curve, 1000 + it, 0, n) if curve != 2 else None
    if curve == 2:
        ks = O.gen_scalars(0, 5000 + it, 0, n)
        dk = ctx.upload(ks); db = ctx.empty((n, 16)); di = ctx.empty((n,), dtype=np.uint8)
        ctx.fixed_base_mul(K.KG_G2, dk.ptr, n, db.ptr, di.ptr); ctx.sync()
        bases = ctx.download(db); binf = ctx.download(di)
    scal = O.gen_scalars(fld, 2000 + it, 0, n)
    pat = int(rng.integers(0, 7))
    if pat == 0: scal[:] = scal[0]                                   # all equal
    elif pat == 1: scal[rng.random(n) < 0.9] = scal[min(1, n - 1)]               # one dominant value among uniform ones
    elif pat == 2:                                                   # few distinct values
        vals = scal[:5].copy(); scal[:] = vals[rng.integers(0, len(vals), n)]
    elif pat == 3 and fld == 0: syn.witness_like(scal, it)
    elif pat == 4: scal[rng.random(n) < 0.5] = 0                     # half zeros
    elif pat == 5: scal[rng.random(n) < 0.97] = scal[min(2, n - 1)]              # 97 % one value
    # pat 6: uniform
    cv = names[curve]
    if curve == 2:
        want = O.to_affine(cv, O.msm(cv, bases, scal, binf, threads=8))
        d_b, d_i = db, di
        d_s = ctx.upload(scal)                      # (kept alive: a temporary would be freed before the call ran)
        got = ctx.msm(curve, d_b.ptr, d_i.ptr, d_s.ptr, n)
    else:
        want = O.to_affine(cv, O.msm(cv, bases, scal, None, threads=8))
        d_b = ctx.upload(bases)
        d_s = ctx.upload(scal)
        got = ctx.msm(curve, d_b.ptr, 0, d_s.ptr, n)
        ctx.msm_begin(curve, d_b.ptr, 0, d_s.ptr, n, 0); got2 = ctx.msm_end(curve, 0)
        if not (got2 == got).all(): bad += 1; print("PIPELINED != BLOCKING", it, cv, n, pat)
    # the same pairs with the scalars in host memory (index slices under the uploads), bases plain and registered
    inf_ptr = d_i.ptr if curve == 2 else 0
    got3 = ctx.msm_host_scalars(curve, d_b.ptr, inf_ptr, scal, n)
    ctx.bases_register(curve, d_b.ptr, inf_ptr, n)
    got4 = ctx.msm_host_scalars(curve, d_b.ptr, inf_ptr, scal, n)
    ctx.bases_unregister(d_b.ptr)
    if not ((got3 == got).all() and (got4 == got).all()): bad += 1; print("HOST SCALARS != BLOCKING", it, cv, n, pat)
    w = 8 if curve != 2 else 16
    ok = (want[1] and not got[w:].any()) or ((not want[1]) and (got[:w] == want[0]).all())
    print(f"{it:3d} {cv} n={n} pattern {pat}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("failures:", bad)
sys.exit(1 if bad else 0)

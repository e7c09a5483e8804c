"""Blocking kg_msm and the four-deep pipelined step over a ladder of lengths (powers of two and the points between), G1, unregistered bases:
cliffs in ms per pair point at thresholds of the automatic choices (window, sort form, window groups, slices).  usage: size_sweep.py [lo hi]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
from kogarashi_amd import lib as LIB
K.init()
gc.disable()         # no cyclic collection inside a timed loop (a 35 ms pause: tools/dbg/anom_1024.py)
SEED = 0x4B6F676172617368
ctx = K.Context(0)
ctx.set_inputs_complete(True)
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 24)
CV = {"g1": (K.KG_G1, K.KG_FR, 8), "gk": (K.KG_GRUMPKIN, K.KG_FQ, 8), "g2": (K.KG_G2, K.KG_FR, 16)}[sys.argv[3] if len(sys.argv) > 3 else "g1"]
CURVE = CV[0]
nmax = 1 << hi
db = ctx.empty((nmax, CV[2])); ds = ctx.empty((nmax, 4))
ctx.gen_scalars(CV[1], SEED + 1, 0, nmax, ds.ptr)
if CURVE == K.KG_G2:                                   # G2 bases: k_i * G2 (no try-and-increment generator for the twist)
    dinf = ctx.empty((nmax,), dtype=np.uint8)
    ctx.fixed_base_mul(2, ds.ptr, nmax, db.ptr, dinf.ptr)
    ctx.gen_scalars(CV[1], SEED + 2, 0, nmax, ds.ptr)
else:
    ctx.gen_bases(CURVE, SEED, 0, nmax, db.ptr)
ctx.sync()
def piped(n, steps):
    for i in range(steps):
        ctx.msm_begin(CURVE, db.ptr, 0, ds.ptr, n, i % 4)
        if i >= 3: ctx.msm_end(CURVE, (i - 3) % 4)
    for i in range(steps - 3, steps): ctx.msm_end(CURVE, i % 4)
piped(min(nmax, 1 << 20), 200 if hi >= 20 else 2000)
for k in range(lo, hi + 1):
    for n in ([1 << k, (1 << k) + 1, 3 << (k - 1)] if k < hi else [1 << k]):
        reps = max(3, min(30, (64 << 20) // n))
        for _ in range(2): ctx.msm(CURVE, db.ptr, 0, ds.ptr, n)
        t0 = time.perf_counter()
        for _ in range(reps): ctx.msm(CURVE, db.ptr, 0, ds.ptr, n)
        b = (time.perf_counter() - t0) / reps * 1e3
        piped(n, 4)
        t0 = time.perf_counter(); piped(n, reps + 3); p = (time.perf_counter() - t0) / (reps + 3) * 1e3
        print(f"n = {n:9d} (2^{np.log2(n):6.3f})  c = {LIB.msm_pick_window(n):>2}  blocking {b:8.3f} ms  {b * 1e6 / n:8.2f} ns/pair   pipelined {p:8.3f} ms  {p * 1e6 / n:8.2f} ns/pair", flush=True)

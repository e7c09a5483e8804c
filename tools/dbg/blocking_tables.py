"""Blocking kg_msm against registered bases, without and with window tables (kg_bases_precompute), and the four-deep pipelined step of both.
usage: blocking_tables.py [log_n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
ctx = K.Context(0)
ctx.set_inputs_complete(True)
sizes = [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20]
def t(f, reps):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r
def piped(curve, db, ds, n, steps):
    for i in range(steps):
        ctx.msm_begin(curve, db.ptr, 0, ds.ptr, n, i % 4)
        if i >= 3: ctx.msm_end(curve, (i - 3) % 4)
    for i in range(steps - 3, steps): ctx.msm_end(curve, i % 4)
for lg in sizes:
    n = 1 << lg
    curve, field = K.KG_G1, K.KG_FR
    db = ctx.empty((n, 8)); ds = ctx.empty((n, 4))
    ctx.gen_bases(curve, SEED, 0, n, db.ptr); ctx.gen_scalars(field, SEED + 1, 0, n, ds.ptr); ctx.sync()
    ctx.bases_register(curve, db.ptr, 0, n)
    piped(curve, db, ds, n, max(8, (200 << 20) // n))
    out = []
    for tables in (0, 1):
        if tables:
            ctx.bases_precompute(db.ptr); ctx.sync()
        rr = []
        for rnd in range(2):
            a, ra = t(lambda: ctx.msm(curve, db.ptr, 0, ds.ptr, n), 20)
            t0 = time.perf_counter(); piped(curve, db, ds, n, 40); p = (time.perf_counter() - t0) / 40 * 1e3
            rr.append(f"{a:.3f}/{p:.3f}")
        out.append(("tables " if tables else "plain ") + " ".join(rr))
    print(f"g1 2^{lg}: blocking/pipelined ms: " + " | ".join(out), flush=True)
    ctx.bases_unregister(db.ptr)
    del db, ds

"""Blocking kg_msm latency at large sizes (sliced pipeline from 2^21 pairs):  python tools/dbg/big_blocking.py 21 22 24"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
ctx = K.Context(0)
for lg in [int(a) for a in sys.argv[1:]] or [21, 22, 24]:
    n = 1 << lg
    db, ds = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, 100 + lg, 0, n, db.ptr); ctx.gen_scalars(K.KG_FR, 200 + lg, 0, n, ds.ptr); ctx.sync()
    for _ in range(6):
        ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    dt = (time.perf_counter() - t0) / reps
    print(f"2^{lg}: {dt * 1e3:.2f} ms blocking -> {n / dt / 1e6:.0f} Mpairs/s", flush=True)
    del db, ds
ctx.close()

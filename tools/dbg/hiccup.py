"""Per-call wall times of kg_msm_begin / kg_msm_end around a change of length (one-time costs inside a pipeline): hiccup.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
ctx = K.Context(0)
ctx.set_inputs_complete(True)
nmax = 1 << 16
db = ctx.empty((nmax, 8)); ds = ctx.empty((nmax, 4))
ctx.gen_bases(K.KG_G1, 1, 0, nmax, db.ptr); ctx.gen_scalars(K.KG_FR, 2, 0, nmax, ds.ptr); ctx.sync()
def run(n, steps, tag):
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, i % 4)
        t1 = time.perf_counter()
        if i >= 3: ctx.msm_end(K.KG_G1, (i - 3) % 4)
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    for i in range(steps - 3, steps): ctx.msm_end(K.KG_G1, i % 4)
    print(tag, " ".join(f"{a:.2f}+{b:.2f}" for a, b in ts), flush=True)
for _ in range(3): ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, 512)
run(512, 12, "512 a:")
for _ in range(3): ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, 1024)
run(1024, 4, "1024 warm:")
run(1024, 16, "1024 timed:")
run(1024, 16, "1024 again:")
run(512, 8, "512 b:")
run(65536, 16, "65536:")
run(1024, 12, "1024 c:")
run(65536, 12, "65536 b:")

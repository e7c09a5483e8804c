"""Host-scalar MSM (kg_msm_host_scalars) against resident bases beside the resident blocking kg_msm, PCIe-inclusive.
usage: host_scalars.py [log_n ...]   (KG_HOST_SLICES / KG_HOST_FIRST_DIV select the cut)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
ctx = K.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [20, 21, 24]
for lg in sizes:
    n = 1 << lg
    for curve, field, name in ((K.KG_G1, K.KG_FR, "g1"),) + (((K.KG_GRUMPKIN, K.KG_FQ, "grumpkin"),) if lg >= 24 else ()):
        db = ctx.empty((n, 8)); ds = ctx.empty((n, 4))
        ctx.gen_bases(curve, SEED, 0, n, db.ptr); ctx.gen_scalars(field, SEED + 1, 0, n, ds.ptr); ctx.sync()
        ctx.bases_register(curve, db.ptr, 0, n)
        hs = ds.numpy()
        t = time.perf_counter(); ctx.write(ds.ptr, hs); up = time.perf_counter() - t
        for _ in range(10 if lg <= 22 else 2): ref = ctx.msm(curve, db.ptr, 0, ds.ptr, n)
        reps = int(os.environ.get('REPS', 30 if lg <= 22 else 6))
        t = time.perf_counter()
        for _ in range(reps): ref = ctx.msm(curve, db.ptr, 0, ds.ptr, n)
        res = (time.perf_counter() - t) / reps
        for _ in range(5 if lg <= 22 else 2): out = ctx.msm_host_scalars(curve, db.ptr, 0, hs, n)
        t = time.perf_counter()
        for _ in range(reps): out = ctx.msm_host_scalars(curve, db.ptr, 0, hs, n)
        host = (time.perf_counter() - t) / reps
        print(f"{name} 2^{lg}: resident blocking {res*1e3:.3f} ms | host scalars {host*1e3:.3f} ms ({host/res:.3f}x, +{(host-res)*1e3:.3f} ms) | "
              f"plain upload {up*1e3:.3f} ms ({n*32/up/1e9:.1f} GB/s) | same={(out == ref).all()}", flush=True)
        ctx.bases_unregister(db.ptr)
        del db, ds

"""Host-memory check of single entry points in a loop (no torch in the process): register / precompute / unregister, setup, proofs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
ctx = K.Context(0)
def rss_kb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"): return int(ln.split()[1])
n = 1 << 16
db = ctx.empty((n, 8)); ds = ctx.empty((n, 4))
ctx.gen_bases(K.KG_G1, 1, 0, n, db.ptr); ctx.gen_scalars(K.KG_FR, 2, 0, n, ds.ptr); ctx.sync()
hs = ds.numpy()
def loop(name, f, reps):
    for _ in range(reps // 10): f()
    a = rss_kb()
    for _ in range(reps): f()
    b = rss_kb()
    print(f"{name:28s} {reps} reps: RSS {a} -> {b} kB ({(b - a) * 1024 / reps:.0f} B per rep)", flush=True)
def reg():
    ctx.bases_register(K.KG_G1, db.ptr, 0, n); ctx.bases_precompute(db.ptr); ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n); ctx.bases_unregister(db.ptr)
loop("register+tables+msm+unreg", reg, 300)
loop("blocking msm", lambda: ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n), 2000)
loop("host scalars 2^16", lambda: ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, hs, n), 2000)
nb = 1 << 19
db2 = ctx.empty((nb, 8)); ds2 = ctx.empty((nb, 4))
ctx.gen_bases(K.KG_G1, 1, 0, nb, db2.ptr); ctx.gen_scalars(K.KG_FR, 2, 0, nb, ds2.ptr); ctx.sync(); hs2 = ds2.numpy()
loop("host scalars 2^19 (sliced)", lambda: ctx.msm_host_scalars(K.KG_G1, db2.ptr, 0, hs2, nb), 1000)
def tick():
    for i in range(4): ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, i)
    for i in range(4): ctx.msm_end(K.KG_G1, i)
loop("4 tickets", tick, 500)
v = ctx.empty((1 << 16, 4)); ctx.gen_scalars(K.KG_FR, 3, 0, 1 << 16, v.ptr)
loop("ntt 2^16", lambda: (ctx.ntt(v.ptr, 16, False, False), ctx.sync()), 2000)
loop("empty/free", lambda: ctx.empty((12345,)), 5000)
loop("upload/numpy", lambda: ctx.upload(hs[:1000]).numpy(), 3000)

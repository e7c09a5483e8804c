"""Soak: a few minutes of mixed calls on one context -- blocking and in-flight MSMs over a ladder of lengths and curves, host-scalar calls,
transforms, proofs -- watching device memory (kg_mem_info) and the process's resident set for growth, and every result for drift.
usage: soak.py [seconds]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = K.Context(0)
rnd = random.Random(11)
def rss_mb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"): return int(ln.split()[1]) / 1024
nmax = 1 << 20
db = ctx.empty((nmax, 8)); ds = ctx.empty((nmax, 4)); dg = ctx.empty((nmax, 8)); dq = ctx.empty((nmax, 4))
ctx.gen_bases(K.KG_G1, SEED, 0, nmax, db.ptr); ctx.gen_scalars(K.KG_FR, SEED + 1, 0, nmax, ds.ptr)
ctx.gen_bases(K.KG_GRUMPKIN, SEED + 2, 0, nmax, dg.ptr); ctx.gen_scalars(K.KG_FQ, SEED + 3, 0, nmax, dq.ptr)
ctx.sync()
hs = ds.numpy()
ctx.bases_register(K.KG_G1, db.ptr, 0, nmax)
sizes = [1, 17, 300, 1 << 10, 5000, 1 << 14, 70000, 1 << 17, 300000, 1 << 19, 1 << 20]
want = {}
def check(key, got):
    got = np.asarray(got).copy()
    if key in want: assert (want[key] == got).all(), key
    else: want[key] = got
nt = ctx.empty((1 << 18, 4)); ctx.gen_scalars(K.KG_FR, SEED + 9, 0, 1 << 18, nt.ptr); ctx.sync(); nt0 = nt.numpy().copy()
t0 = time.time(); it = 0; last = 0
base = None
while time.time() - t0 < secs:
    n = rnd.choice(sizes)
    kind = rnd.randrange(6)
    if kind == 0: check(("g1", n), ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n))
    elif kind == 1: check(("gk", n), ctx.msm(K.KG_GRUMPKIN, dg.ptr, 0, dq.ptr, n))
    elif kind == 2: check(("g1", n), ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, hs[:n], n))
    elif kind == 3:
        for i in range(8):
            ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, i % 4)
            if i >= 3: check(("g1", n), ctx.msm_end(K.KG_G1, (i - 3) % 4))
        for i in range(5, 8): check(("g1", n), ctx.msm_end(K.KG_G1, i % 4))
    elif kind == 4:
        lg = rnd.choice([6, 10, 14, 18])
        ctx.ntt(nt.ptr, lg, False, False); ctx.ntt(nt.ptr, lg, True, False); ctx.sync()
        assert (nt.numpy()[: 1 << lg] == nt0[: 1 << lg]).all()
    else:
        xy, inf = ctx.commit_host_scalars(K.KG_G1, db.ptr, 0, hs[:n], n); check(("c", n), xy)
        b = ctx.empty((rnd.randrange(1, 1 << 21),)); del b          # pooled kg_malloc / kg_free of odd sizes
    it += 1
    if time.time() - t0 - last >= 10:
        last = time.time() - t0
        fr, tot = ctx.mem_info()
        cur = (round(rss_mb()), round((tot - fr) / 2**20))
        if it > 200 and base is None: base = cur
        print(f"t = {last:5.0f} s  calls {it:6d}  host RSS {cur[0]} MiB  device in use {cur[1]} MiB", flush=True)
fr, tot = ctx.mem_info()
end = (round(rss_mb()), round((tot - fr) / 2**20))
print("base", base, "end", end, flush=True)
assert base is None or (end[0] - base[0] < 256 and end[1] - base[1] < 1024), "memory grew over the soak"
print("soak ok:", it, "calls,", len(want), "distinct results, all stable")

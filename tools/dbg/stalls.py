"""Rare long calls in a pipelined loop of short MSMs: per-call wall times of kg_msm_begin / kg_msm_end over many iterations, outliers listed.
   python tools/dbg/stalls.py [n] [iterations]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
gc.disable()
ctx = K.Context(0); ctx.set_inputs_complete(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
db = ctx.empty((n, 8)); ds = ctx.empty((n, 4))
ctx.gen_scalars(0, 1, 0, n, ds.ptr); ctx.gen_bases(0, 2, 0, n, db.ptr); ctx.sync()
for i in range(8):
    ctx.msm_begin(0, db.ptr, 0, ds.ptr, n, i % 4)
    if i >= 3: ctx.msm_end(0, (i - 3) % 4)
for i in range(5, 8): ctx.msm_end(0, i % 4)
tb, te = np.zeros(iters), np.zeros(iters)
t_start = time.perf_counter()
for i in range(iters):
    t0 = time.perf_counter(); ctx.msm_begin(0, db.ptr, 0, ds.ptr, n, i % 4); t1 = time.perf_counter()
    if i >= 3: ctx.msm_end(0, (i - 3) % 4)
    t2 = time.perf_counter()
    tb[i], te[i] = (t1 - t0) * 1e6, (t2 - t1) * 1e6
total = time.perf_counter() - t_start
for i in range(iters - 3, iters): ctx.msm_end(0, i % 4)
print(f"n = {n}: {iters} calls in {total * 1e3:.1f} ms ({total / iters * 1e6:.1f} us per call); begin median {np.median(tb):.1f} us, end median {np.median(te):.1f} us")
for name, arr in (("begin", tb), ("end", te)):
    idx = np.nonzero(arr > 1000.0)[0]
    print(f"  {name}: {len(idx)} calls over 1 ms:", " ".join(f"#{i}:{arr[i] / 1e3:.2f}ms" for i in idx[:30]))
    print(f"  {name}: percentiles 50/99/99.9/max = {np.percentile(arr, 50):.0f} / {np.percentile(arr, 99):.0f} / {np.percentile(arr, 99.9):.0f} / {arr.max():.0f} us")

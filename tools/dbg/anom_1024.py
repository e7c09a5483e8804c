import os, sys, time, gc
if os.environ.get('NOGC'): gc.disable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
ctx = K.Context(0); ctx.set_inputs_complete(True)
hi = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nmax = 1 << hi
db = ctx.empty((nmax, 8)); ds = ctx.empty((nmax, 4))
ctx.gen_scalars(0, 1, 0, nmax, ds.ptr); ctx.gen_bases(0, 2, 0, nmax, db.ptr); ctx.sync()
def piped(n, steps, rec=None):
    for i in range(steps):
        t0 = time.perf_counter(); ctx.msm_begin(0, db.ptr, 0, ds.ptr, n, i % 4); t1 = time.perf_counter()
        if i >= 3: ctx.msm_end(0, (i - 3) % 4)
        t2 = time.perf_counter()
        if rec is not None: rec.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6))
    for i in range(steps - 3, steps):
        t0 = time.perf_counter(); ctx.msm_end(0, i % 4)
        if rec is not None: rec.append((-1, (time.perf_counter() - t0) * 1e6))
piped(min(nmax, 1 << 20), 200)
gc.disable()
for k in range(4, 14):
  for n in [1 << k, (1 << k) + 1, 3 << (k - 1)]:
    reps = max(3, min(30, (64 << 20) // n))
    for _ in range(2): ctx.msm(0, db.ptr, 0, ds.ptr, n)
    bt = []
    for _ in range(reps):
        t0 = time.perf_counter(); ctx.msm(0, db.ptr, 0, ds.ptr, n); bt.append((time.perf_counter() - t0) * 1e6)
    piped(n, 4)
    rec = []
    t0 = time.perf_counter(); piped(n, reps + 3, rec); p = (time.perf_counter() - t0) / (reps + 3) * 1e3
    if n >= 4096: print(n, f"blocking {sum(bt) / len(bt) / 1e3:.3f} ms:", " ".join(f"{v:.0f}" for v in bt), f"\n   pipelined {p:.3f} ms", " ".join(f"{a:.0f}/{b:.0f}" for a, b in rec))

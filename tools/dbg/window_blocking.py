"""Blocking kg_msm latency against the window width at small sizes (the per-rank unit of a strong-scaled MSM): python tools/dbg/window_blocking.py lg c..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kogarashi_amd as K
K.init()
ctx = K.Context(0)
ctx.set_inputs_complete(True)
import numpy as np
lg = int(sys.argv[1]); n = 1 << lg
CV = os.environ.get("CURVE", "g1")
CURVE = {"g1": K.KG_G1, "g2": K.KG_G2}[CV]
b, s = ctx.empty((n, 16 if CV == "g2" else 8)), ctx.empty((n, 4))
ctx.gen_scalars(K.KG_FR, 2, 0, n, s.ptr)
if CV == "g2":
    di = ctx.empty((n,), dtype=np.uint8)
    ctx.fixed_base_mul(2, s.ptr, n, b.ptr, di.ptr); ctx.gen_scalars(K.KG_FR, 3, 0, n, s.ptr)
else:
    ctx.gen_bases(K.KG_G1, 1, 0, n, b.ptr)
ctx.sync()
ctx.bases_register(CURVE, b.ptr, 0, n)
ref = None
for i in range(300): ctx.msm_begin(CURVE, b.ptr, 0, s.ptr, n, i % 4); (i >= 3) and ctx.msm_end(CURVE, (i - 3) % 4)
for i in range(297, 300): ctx.msm_end(CURVE, i % 4)
for c in [0] + [int(a) for a in sys.argv[2:]] + [0]:
    ctx.set_msm_window(c)
    for _ in range(5): r = ctx.msm(CURVE, b.ptr, 0, s.ptr, n)
    t = time.perf_counter()
    for _ in range(30): r = ctx.msm(CURVE, b.ptr, 0, s.ptr, n)
    dt = (time.perf_counter() - t) / 30 * 1e3
    ref = r if ref is None else ref
    print(f"2^{lg} c={c or 'auto(%d)' % K.lib.msm_pick_window(n)}: {dt:.3f} ms  same={(r == ref).all()}", flush=True)

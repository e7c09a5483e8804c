"""kg_fixed_base_mul time per 2^18 vector, G1 and G2:  python tools/dbg/fixed_base_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
ctx = K.Context(0)
n = 1 << 18
dk = ctx.empty((n, 4)); ctx.gen_scalars(K.KG_FR, 7, 0, n, dk.ptr)
for curve, w in ((K.KG_G1, 8), (K.KG_G2, 16)):
    dxy, dinf = ctx.empty((n, w)), ctx.empty((n,), dtype=np.uint8)
    ctx.fixed_base_mul(curve, dk.ptr, n, dxy.ptr, dinf.ptr); ctx.sync()
    t = time.perf_counter()
    for _ in range(5): ctx.fixed_base_mul(curve, dk.ptr, n, dxy.ptr, dinf.ptr)
    ctx.sync()
    print(f"curve {curve}: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per 2^18 multiples")

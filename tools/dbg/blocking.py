"""Blocking-call latencies: kg_msm at 2^20 / 2^18, kg_commit, kg_ntt:  python tools/dbg/blocking.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
ctx = K.Context(0)
for mode in ("stream-ordered", "inputs complete"):
    ctx.set_inputs_complete(mode != "stream-ordered")
    for lg in (20, 18, 14):
        n = 1 << lg
        b, s = ctx.empty((n, 8)), ctx.empty((n, 4))
        ctx.gen_bases(K.KG_G1, 1, 0, n, b.ptr); ctx.gen_scalars(K.KG_FR, 2, 0, n, s.ptr); ctx.sync()
        for _ in range(3): ctx.msm(K.KG_G1, b.ptr, 0, s.ptr, n)
        t = time.perf_counter()
        for _ in range(10): ctx.msm(K.KG_G1, b.ptr, 0, s.ptr, n)
        print(f"{mode:16s} kg_msm 2^{lg}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms", flush=True)

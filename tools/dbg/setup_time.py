"""Device CRS setup (api.groth16_setup) wall time at 2^16 / 2^18 constraints:  python tools/dbg/setup_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
from kogarashi_amd import synthetic as syn
from kogarashi_amd.api import groth16_setup
ctx = K.Context(0)
for lg in (16, 18):
    m = 1 << lg
    cc = syn.ChainCircuit(m)
    for rep in range(2):
        t0 = time.perf_counter()
        ctx.profile_enable(True)
        P = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=ctx)
        ctx.sync()
        print(f"setup 2^{lg} rep {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms", {k: (round(v[0], 2), v[1]) for k, v in ctx.profile_summary().items()})
        ctx.profile_enable(False)

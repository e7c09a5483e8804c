"""Blocking kg_commit of 2^lg registered pairs (KG_MSM_GROUPS etc. from the environment): python tools/dbg/commit24.py [lg] [witness]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
ctx = K.Context(0)
ctx.set_inputs_complete(True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
ctx.gen_bases(K.KG_G1, SEED + 40, 0, n, g.ptr); ctx.gen_scalars(K.KG_FR, SEED + 41, 0, n, m.ptr); ctx.sync()
if len(sys.argv) > 2:
    from kogarashi_amd import synthetic as syn
    hm = m.numpy(); syn.witness_like(hm, 23); ctx.write(m.ptr, hm)
ctx.bases_register(K.KG_G1, g.ptr, 0, n)
if os.environ.get('KG_FORCE_C'): ctx.set_msm_window(int(os.environ['KG_FORCE_C']))
for _ in range(4): r = ctx.commit(K.KG_G1, g.ptr, 0, m.ptr, n)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(5): r = ctx.commit(K.KG_G1, g.ptr, 0, m.ptr, n)
    ts.append((time.perf_counter() - t0) / 5 * 1e3)
print(f"2^{lg} commit KG_MSM_GROUPS={os.environ.get('KG_MSM_GROUPS', 'auto')} KG_WIDE_WINDOW={os.environ.get('KG_WIDE_WINDOW', '-')} c={os.environ.get('KG_FORCE_C', 'auto')}: " + " ".join(f"{t:.2f}" for t in ts) + f" ms  x0={int(r[0][0]):016x}", flush=True)

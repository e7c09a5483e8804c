"""Device CRS setup (kg_groth16_setup_bn254 through api.groth16_setup) at 2^16 / 2^18 constraints, with a kernel-level breakdown when run under
rocprofv3 --kernel-trace --stats:  python tools/dbg/setup_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
from kogarashi_amd import synthetic as syn
from kogarashi_amd.lib import Groth16Crs
ctx = K.Context(0)
for lg in (16, 18):
    m = 1 << lg
    cc = syn.ChainCircuit(m)
    nv = cc.l + cc.m_l_1
    mats, keep = [], []
    for rp, col, val in (cc.a, cc.b, cc.c):
        d = [ctx.upload(np.ascontiguousarray(rp, dtype=np.uint64)), ctx.upload(np.ascontiguousarray(col, dtype=np.uint64)),
             ctx.upload(np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4))]
        keep.append(d); mats.append(tuple(x.ptr for x in d))
    lens = {"h": (m - 1, 8), "l": (cc.m_l_1, 8), "a": (nv, 8), "b_g1": (nv, 8), "b_g2": (nv, 16), "ic": (cc.l, 8)}
    dev = {k: (ctx.empty((max(c, 1), w)), ctx.empty((max(c, 1),), dtype=np.uint8)) for k, (c, w) in lens.items()}
    crs = Groth16Crs()
    for k in ("h", "l", "a", "b_g1", "b_g2"):
        setattr(crs, "d_" + k, dev[k][0].ptr); setattr(crs, "d_" + k + "_inf", dev[k][1].ptr)
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.groth16_setup(mats[0], mats[1], mats[2], m, cc.l, cc.m_l_1, syn.fixed_toxic(), crs, dev["ic"][0].ptr, dev["ic"][1].ptr)
        print(f"kg_groth16_setup_bn254 2^{lg} rep {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms (device arrays in and out)", flush=True)

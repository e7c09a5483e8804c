"""Host-memory check of the prover's life cycle without torch: setup -> Prover (registered CRS, window tables) -> proofs -> drop."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
from kogarashi_amd import synthetic as syn
from kogarashi_amd.api import groth16_setup, Prover
ctx = K.Context(0)
def rss_kb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"): return int(ln.split()[1])
m = 1 << 12
cc = syn.ChainCircuit(m)
a_csr, b_csr, c_csr = cc.a, cc.b, cc.c
l, m_l_1 = len(cc.x), len(cc.w)
toxic = np.arange(1, 21, dtype=np.uint64).reshape(5, 4)
ctx_up = ctx.upload(toxic); ctx.field_vec_op(K.KG_FR, "to_mont", ctx_up.ptr, 0, ctx_up.ptr, 5); toxic = ctx_up.numpy()
ev = cc.evaluate() if hasattr(cc, "evaluate") else None
r = toxic[0]; s = toxic[1]
def cycle(tables):
    P = groth16_setup(a_csr, b_csr, c_csr, m, l, m_l_1, toxic, None, ctx=ctx)
    pr = Prover(P, m, l, m_l_1, ctx=ctx, window_tables=tables)
    pr.attach_constraint_system(a_csr, b_csr, c_csr)
    for _ in range(3):
        pr.create_proof_from_witness(cc.x, cc.w, r, s)
    del pr, P
for name, tb, reps in (("setup+prover+3 proofs", False, 60), ("same with tables", True, 60)):
    for _ in range(6): cycle(tb)
    gc.collect(); a = rss_kb()
    for _ in range(reps): cycle(tb)
    gc.collect(); b = rss_kb()
    fr, tot = ctx.mem_info()
    print(f"{name:26s} {reps} cycles: RSS {a} -> {b} kB ({(b - a) / reps:.1f} kB per cycle), device in use {(tot - fr) >> 20} MiB", flush=True)

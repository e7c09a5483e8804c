"""kg_nova_cross_term and kg_r1cs_prod on the chain circuit at m = 2^lg constraints: time, and the HBM rate of its algorithmic traffic
(40 B per non-zero: column + coefficient; 32 B per entry of z1 and z2; 32 B per output row):  python tools/dbg/nova_rate.py 18 20"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
from kogarashi_amd import synthetic as syn
ctx = K.Context(0)
for lg in [int(a) for a in sys.argv[1:]] or [18, 20]:
    m = 1 << lg
    cc = syn.ChainCircuit(m)
    shape = (cc.a, cc.b, cc.c)
    nnz = sum(len(t[1]) for t in shape)
    dev = [tuple(ctx.upload(np.ascontiguousarray(x)) for x in (rp, col, np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4))) for rp, col, val in shape]
    z = np.concatenate([cc.x, cc.w])
    z1, z2 = ctx.upload(z), ctx.upload(z[::-1].copy())
    out = ctx.empty((m, 4))
    ptrs = [tuple(d.ptr for d in trip) for trip in dev]
    u = np.ascontiguousarray(cc.x[0])
    for _ in range(3):
        ctx.nova_cross_term(K.KG_FR, ptrs[0], ptrs[1], ptrs[2], m, z1.ptr, z2.ptr, u, u, out.ptr)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        ctx.nova_cross_term(K.KG_FR, ptrs[0], ptrs[1], ptrs[2], m, z1.ptr, z2.ptr, u, u, out.ptr)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    by = nnz * 40 + 2 * len(z) * 32 + m * 32
    print(f"m = 2^{lg}: nnz {nnz}, {dt * 1e6:.1f} us per cross term, {by / dt / 1e9:.0f} GB/s of algorithmic traffic ({by / 1e6:.1f} MB)", flush=True)
    # one matrix-vector product of the same shape (kg_r1cs_prod: what cs.evaluate() runs three times)
    for _ in range(3):
        ctx.r1cs_prod(K.KG_FR, ptrs[0][0], ptrs[0][1], ptrs[0][2], m, z1.ptr, out.ptr)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.r1cs_prod(K.KG_FR, ptrs[0][0], ptrs[0][1], ptrs[0][2], m, z1.ptr, out.ptr)
    ctx.sync()
    dp = (time.perf_counter() - t0) / reps
    bp = len(shape[0][1]) * 40 + len(z) * 32 + m * 32
    print(f"          A z: {dp * 1e6:.1f} us, {bp / dp / 1e9:.0f} GB/s", flush=True)
ctx.close()

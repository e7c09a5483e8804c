"""Groth16 2^18, pipelined only (no tables leg): python tools/dbg/g16_plain.py [tickets] [tables 0/1] -- for kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
from kogarashi_amd import synthetic as syn
from kogarashi_amd.api import groth16_setup
tickets = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tables = len(sys.argv) > 2 and sys.argv[2] == "1"
ctx = K.Context(0)
ctx.set_inputs_complete(True)
m = 1 << 18
cc = syn.ChainCircuit(m)
P = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=ctx)
r, s = syn.fixed_rs()
prover = K.Prover(P, m, cc.l, cc.m_l_1, ctx=ctx, window_tables=tables)
up = lambda v: ctx.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
d = [up(v) for v in (cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w)]
args = (prover.crs, *[x.ptr for x in d], r, s)
def run(k, depth=tickets):
    last = None
    for i in range(k):
        ctx.groth16_prove_begin(*args, i % depth)
        if i >= depth - 1:
            last = ctx.groth16_prove_end((i - depth + 1) % depth)
    for i in range(max(k - depth + 1, 0), k):
        last = ctx.groth16_prove_end(i % depth)
    return last
run(6)
ctx.sync()
t0 = time.perf_counter()
run(20)
ctx.sync()
print(f"tickets {tickets} tables {tables}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per proof")
ctx.close()

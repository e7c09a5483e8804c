import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
from oracle import oracle as O
ctx = K.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = O.generator("g2")
one = np.concatenate([O.f_consts(1)["r"], np.zeros(4, dtype=np.uint64)])
gen_proj = np.concatenate([g, one])
ks = O.gen_scalars(0, 77, 0, n)
bases = np.stack([O.to_affine("g2", O.scalar_point("g2", gen_proj, ks[i]))[0] for i in range(n)])
scal = O.gen_scalars(0, 78, 0, n)
if len(sys.argv) > 2: scal[:] = O.f_consts(0)["r"]     # all ones
want = O.to_affine("g2", O.msm("g2", bases, scal, None, threads=4))
out = ctx.msm_host(2, bases, None, scal, n)
print("want", [hex(int(x)) for x in want[0][:4]])
print("got ", [hex(int(x)) for x in out[:4]], "z", out[16:24])
print("eq words", (out[:16] == want[0]).astype(int)); print("want c1", [hex(int(x)) for x in want[0][4:8]]); print("got  c1", [hex(int(x)) for x in out[4:8]])

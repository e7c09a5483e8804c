"""Pipelined MSM step time (4 in flight, as bench.py) against the window width c:  python tools/dbg/window_pipe.py <log_n> c1 c2 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
ctx.set_inputs_complete(True)
lg = int(sys.argv[1]); n = 1 << lg
bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_bases(K.KG_G1, SEED + lg, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + lg + 1, 0, n, scal.data_ptr()); ctx.sync()
depth = 4
def run(k):
    out = None
    for i in range(k):
        ctx.msm_begin(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n, i % 4)
        if i >= depth - 1: out = ctx.msm_end(K.KG_G1, (i - depth + 1) % 4)
    for i in range(max(k - depth + 1, 0), k): out = ctx.msm_end(K.KG_G1, i % 4)
    return out
ref = None
for rep in range(2):
    for c in [int(a) for a in sys.argv[2:]]:
        ctx.set_msm_window(c)
        r = run(8)
        t = time.time(); r = run(40); dt = (time.time() - t) / 40
        if ref is None: ref = r
        print(f"2^{lg} c={c}: {dt*1e3:.3f} ms/step same={(r == ref).all()}", flush=True)

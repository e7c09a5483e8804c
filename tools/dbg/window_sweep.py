"""MSM time against the window width c at large n:  python tools/dbg/window_sweep.py <log_n> c1 c2 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
lg = int(sys.argv[1]); n = 1 << lg
bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_bases(K.KG_G1, SEED + lg, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + lg + 1, 0, n, scal.data_ptr()); ctx.sync()
ref = None
for c in [int(a) for a in sys.argv[2:]]:
    ctx.set_msm_window(c)
    r = ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n)
    ts = []
    for _ in range(3):
        t = time.time(); r = ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n); ts.append(time.time() - t)
    if ref is None: ref = r
    print(f"2^{lg} c={c}: {min(ts)*1e3:.2f} ms  {n/min(ts)/1e6:.1f} Mpairs/s same={(r == ref).all()}", flush=True)

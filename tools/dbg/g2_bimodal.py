"""Blocking G2 MSM of a fixed short length, timed in batches of 10: is the time bimodal?  python tools/dbg/g2_bimodal.py [n] [batches]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import kogarashi_amd as K
K.init(); gc.disable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
ctx = K.Context(0); ctx.set_inputs_complete(True)
s = torch.empty(n * 4, dtype=torch.int64, device=dev); ctx.gen_scalars(0, 77, 0, n, s.data_ptr())
b = torch.empty(n * 16, dtype=torch.int64, device=dev); inf = torch.zeros(n, dtype=torch.uint8, device=dev)
ctx.fixed_base_mul(2, s.data_ptr(), n, b.data_ptr(), inf.data_ptr()); ctx.gen_scalars(0, 78, 0, n, s.data_ptr()); ctx.sync()
for _ in range(5): ctx.msm(2, b.data_ptr(), inf.data_ptr(), s.data_ptr(), n)
out = []
for _ in range(batches):
    t0 = time.perf_counter()
    for _ in range(10): ctx.msm(2, b.data_ptr(), inf.data_ptr(), s.data_ptr(), n)
    out.append((time.perf_counter() - t0) / 10 * 1e3)
print(f"n = {n}:", " ".join(f"{v:.3f}" for v in out))
ctx.profile_enable(True)
for _ in range(50): ctx.msm(2, b.data_ptr(), inf.data_ptr(), s.data_ptr(), n)
sm = ctx.profile_summary()
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in sm.items()}, "us")

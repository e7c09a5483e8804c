"""Blocking G2 MSM phases at 2^lg (bases = k_i * G2 from the device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
lg = int(sys.argv[1]); n = 1 << lg
k = torch.empty(n * 4, dtype=torch.int64, device=dev); ctx.gen_scalars(K.KG_FR, SEED + 5, 0, n, k.data_ptr())
bases = torch.empty(n * 16, dtype=torch.int64, device=dev); inf = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.fixed_base_mul(K.KG_G2, k.data_ptr(), n, bases.data_ptr(), inf.data_ptr())
scal = torch.empty(n * 4, dtype=torch.int64, device=dev); ctx.gen_scalars(K.KG_FR, SEED + 6, 0, n, scal.data_ptr()); ctx.sync()
for _ in range(2): r = ctx.msm(K.KG_G2, bases.data_ptr(), 0, scal.data_ptr(), n)
ctx.profile_enable(True)
for _ in range(3): r = ctx.msm(K.KG_G2, bases.data_ptr(), 0, scal.data_ptr(), n)
s = ctx.profile_summary(); ctx.profile_enable(False)
print(f"G2 2^{lg}:", {k_: round(v[0] / v[1], 3) for k_, v in s.items()}, hex(int(r[0])))

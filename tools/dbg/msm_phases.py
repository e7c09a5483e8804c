"""Blocking kg_msm phase times (HIP events) at a given size: python tools/dbg/msm_phases.py lg [c]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
lg = int(sys.argv[1]); n = 1 << lg
if len(sys.argv) > 2: ctx.set_msm_window(int(sys.argv[2]))
bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_bases(K.KG_G1, SEED + lg, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + lg + 1, 0, n, scal.data_ptr()); ctx.sync()
for _ in range(2): ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n)
ctx.profile_enable(True)
for _ in range(3): ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n)
s = ctx.profile_summary(); ctx.profile_enable(False)
print(f"2^{lg}:", {k: round(v[0] / v[1], 3) for k, v in s.items()})

"""kg_msm_host (host buffers in, PCIe inclusive): msm_host_rate.py [lg ...]   (KG_HOST_SLICES selects the slice count)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
for lg in [int(a) for a in sys.argv[1:]] or [20]:
  n = 1 << lg
  bases = torch.empty(n * 8, dtype=torch.int64, device=dev); scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
  ctx.gen_bases(K.KG_G1, SEED, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + 1, 0, n, scal.data_ptr()); ctx.sync()
  hb = bases.cpu().numpy().view(np.uint64).reshape(n, 8); hs = scal.cpu().numpy().view(np.uint64).reshape(n, 4)
  ref = ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n)
  for _ in range(2): out = ctx.msm_host(K.KG_G1, hb, None, hs, n)
  t = time.perf_counter()
  for _ in range(10): out = ctx.msm_host(K.KG_G1, hb, None, hs, n)
  dt = (time.perf_counter() - t) / 10
  print(f"kg_msm_host 2^{lg} KG_HOST_SLICES={os.environ.get('KG_HOST_SLICES', 'auto')}: {dt*1e3:.2f} ms  {n/dt/1e6:.0f} Mpairs/s  same={(out == ref).all()}")

"""Accumulate-phase rate (G mixed additions/s) against n for a forced window width, the accumulation running alone (one window
group, no index slices):  python tools/dbg/acc_rate.py c lg1 lg2 ..."""
import os, sys
os.environ.setdefault("KG_MSM_SLICED", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
ctx.set_msm_groups(1)
c = int(sys.argv[1])
for lg in [int(a) for a in sys.argv[2:]]:
    n = 1 << lg
    bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
    scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_bases(K.KG_G1, SEED + lg, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + lg + 1, 0, n, scal.data_ptr()); ctx.sync()
    ctx.set_msm_window(c)
    for _ in range(3): ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n)
    ctx.profile_enable(True)
    for _ in range(5): ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n)
    s = ctx.profile_summary(); ctx.profile_enable(False)
    acc = s["accumulate"][0] / s["accumulate"][1]
    W = (255 + c - 1) // c
    print(f"c={c} n=2^{lg}: accumulate {acc*1e3:.0f} us, {W*n/acc/1e6:.2f} G madd/s, avg bucket {n/(1<<(c-1)):.1f}, waves {W*(1<<(c-1))//64}", flush=True)

"""Phase times inside the short-input kernel (A/B build, KG_SMALL_STAMPS=1): python tools/dbg/small_stamps.py n c r [curve]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["KG_LIB_PATH"] = os.path.join(root, "kogarashi_amd", "libkogarashi_amd_exp.so")
os.environ["KG_SMALL_STAMPS"] = "1"
import torch
import kogarashi_amd as K
n, c, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
curve = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
ctx = K.Context(0)
s = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_scalars(1 if curve == 1 else 0, 77, 0, n, s.data_ptr())
ip = 0
if curve == 2:                      # G2 bases: generator multiples
    b = torch.empty(n * 16, dtype=torch.int64, device=dev)
    inf = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.fixed_base_mul(2, s.data_ptr(), n, b.data_ptr(), inf.data_ptr())
    ctx.gen_scalars(0, 78, 0, n, s.data_ptr())
    ip = inf.data_ptr()
else:
    b = torch.empty(n * 8, dtype=torch.int64, device=dev)
    ctx.gen_bases(curve, 76, 0, n, b.data_ptr())
ctx.sync()
ctx.set_msm_small(32768, c, r)
print(f"n = {n} c = {c} r = {r}", file=sys.stderr)
for _ in range(6):
    ctx.msm(curve, b.data_ptr(), ip, s.data_ptr(), n)

"""Phase timeline (library HIP events) of a few pipelined 2^20 MSMs: KG_PROFILE_TIMELINE=1 python tools/dbg/msm_timeline.py [depth] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
if os.environ.get("KG_ORDERED") != "1":
    ctx.set_inputs_complete(True)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << 20
bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_bases(K.KG_G1, SEED + 1, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + 2, 0, n, scal.data_ptr()); ctx.sync()
def run(k):
    for i in range(k):
        ctx.msm_begin(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n, i % 4)
        if i >= depth - 1:
            ctx.msm_end(K.KG_G1, (i - depth + 1) % 4)
    for i in range(max(k - depth + 1, 0), k):
        ctx.msm_end(K.KG_G1, i % 4)
run(12)
ctx.profile_enable(True)
run(steps)
ctx.profile_summary()

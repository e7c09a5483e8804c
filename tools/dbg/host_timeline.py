"""A/B of the blocking 2^20 MSM (unregistered / registered bases) and the host-scalar entry, interleaved, then one traced host-scalar call.
KG_TRACE_HOST=1 KG_PROFILE_TIMELINE=1 python tools/dbg/host_timeline.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
if os.environ.get("KG_ORDERED") != "1":
    ctx.set_inputs_complete(True)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
bases = torch.empty(n * 8, dtype=torch.int64, device=dev); breg = torch.empty(n * 8, dtype=torch.int64, device=dev)
scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_bases(K.KG_G1, SEED + 1, 0, n, bases.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + 2, 0, n, scal.data_ptr()); ctx.sync()
breg.copy_(bases); torch.cuda.synchronize()
ctx.bases_register(K.KG_G1, breg.data_ptr(), 0, n)
hs = scal.cpu().numpy().view(np.uint64).reshape(n, 4)
def t(f, reps=20):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r
# warm the clocks like the bench does
for i in range(200):
    ctx.msm_begin(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n, i % 4)
    if i >= 3: ctx.msm_end(K.KG_G1, (i - 3) % 4)
for i in range(197, 200): ctx.msm_end(K.KG_G1, i % 4)
for rnd in range(3):
    a, ra = t(lambda: ctx.msm(K.KG_G1, bases.data_ptr(), 0, scal.data_ptr(), n))
    b, rb = t(lambda: ctx.msm(K.KG_G1, breg.data_ptr(), 0, scal.data_ptr(), n))
    c, rc = t(lambda: ctx.msm_host_scalars(K.KG_G1, breg.data_ptr(), 0, hs, n))
    print(f"round {rnd}: blocking unregistered {a:.3f} ms | registered {b:.3f} ms | host scalars {c:.3f} ms (+{c-a:.3f} / +{c-b:.3f}) same={(ra==rb).all() and (rb==rc).all()}", flush=True)
if os.environ.get("KG_PROFILE_TIMELINE"):
    ctx.profile_enable(True)
    ctx.msm_host_scalars(K.KG_G1, breg.data_ptr(), 0, hs, n)
    ctx.profile_summary()
    ctx.profile_enable(False)
    print("---- registered blocking", file=sys.stderr, flush=True)
    ctx.profile_enable(True)
    ctx.msm(K.KG_G1, breg.data_ptr(), 0, scal.data_ptr(), n)
    ctx.profile_summary()

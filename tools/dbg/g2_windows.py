"""G2 MSM against the window width: blocking and four in flight, 2^lg pairs, widths from the command line (0 = automatic).
   python tools/dbg/g2_windows.py 18 0 12 13 14 15 16"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()
gc.disable()
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0); ctx.set_inputs_complete(True)
lg = int(sys.argv[1]); n = 1 << lg
k = torch.empty(n * 4, dtype=torch.int64, device=dev); ctx.gen_scalars(K.KG_FR, SEED + 5, 0, n, k.data_ptr())
bases = torch.empty(n * 16, dtype=torch.int64, device=dev); inf = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.fixed_base_mul(K.KG_G2, k.data_ptr(), n, bases.data_ptr(), inf.data_ptr())
scal = torch.empty(n * 4, dtype=torch.int64, device=dev); ctx.gen_scalars(K.KG_FR, SEED + 6, 0, n, scal.data_ptr()); ctx.sync()
def piped(steps):
    for i in range(steps):
        ctx.msm_begin(K.KG_G2, bases.data_ptr(), inf.data_ptr(), scal.data_ptr(), n, i % 4)
        if i >= 3: ctx.msm_end(K.KG_G2, (i - 3) % 4)
    for i in range(steps - 3, steps): ctx.msm_end(K.KG_G2, i % 4)
ref = None
for c in [int(a) for a in sys.argv[2:]]:
    ctx.set_msm_window(c)
    for _ in range(3): r = ctx.msm(K.KG_G2, bases.data_ptr(), inf.data_ptr(), scal.data_ptr(), n)
    if ref is None: ref = r
    assert (r == ref).all()
    bl, pp = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10): ctx.msm(K.KG_G2, bases.data_ptr(), inf.data_ptr(), scal.data_ptr(), n)
        bl.append((time.perf_counter() - t0) / 10 * 1e3)
    piped(8)
    for _ in range(3):
        t0 = time.perf_counter(); piped(24); pp.append((time.perf_counter() - t0) / 24 * 1e3)
    print(f"2^{lg} c = {c:2d}: blocking {sorted(bl)[1]:.3f} ms   four in flight {sorted(pp)[1]:.3f} ms per MSM", flush=True)

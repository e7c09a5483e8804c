"""Host-scalar MSM (kg_msm_host_scalars) against registered bases beside the resident blocking kg_msm, PCIe-inclusive, clocks warmed like
the bench.  usage: host_scalars.py [log_n ...]   (KG_HOST_SLICES / KG_HOST_FIRST_DIV / KG_HOST_WINDOW_WHOLE / KG_HOST_ACCQ select the variant)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kogarashi_amd as K
K.init()
SEED = 0x4B6F676172617368
ctx = K.Context(0)
ctx.set_inputs_complete(True)
sizes = [int(a) for a in sys.argv[1:]] or [20, 21, 24]
def t(f, reps):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r
for lg in sizes:
    n = 1 << lg
    for curve, field, name in ((K.KG_G1, K.KG_FR, "g1"),):
        db = ctx.empty((n, 8)); ds = ctx.empty((n, 4))
        ctx.gen_bases(curve, SEED, 0, n, db.ptr); ctx.gen_scalars(field, SEED + 1, 0, n, ds.ptr); ctx.sync()
        ctx.bases_register(curve, db.ptr, 0, n)
        if os.environ.get('TABLES') == '1':
            ctx.bases_precompute(db.ptr); ctx.sync()
        hs = ds.numpy()
        if os.environ.get('TORCH') == '1':             # the bench's host array: a torch CPU tensor viewed by numpy
            import torch
            tt = torch.empty(n * 4, dtype=torch.int64, device="cuda:0")
            ctx.gen_scalars(field, SEED + 1, 0, n, tt.data_ptr()); ctx.sync()
            hs = tt.cpu().numpy().view(np.uint64).reshape(n, 4)
        elif os.environ.get('TORCH') == '2':
            import torch                                # torch loaded, numpy array as before
        warm = max(4, (200 << 20) // n)
        for i in range(warm):                       # clock ramp: pipelined MSMs as in the bench's timed region
            ctx.msm_begin(curve, db.ptr, 0, ds.ptr, n, i % 4)
            if i >= 3: ctx.msm_end(curve, (i - 3) % 4)
        for i in range(warm - 3, warm): ctx.msm_end(curve, i % 4)
        reps = int(os.environ.get('REPS', 20 if lg <= 22 else 5))
        out = []
        for rnd in range(2):
            a, ra = t(lambda: ctx.msm(curve, db.ptr, 0, ds.ptr, n), reps)
            c, rc = t(lambda: ctx.msm_host_scalars(curve, db.ptr, 0, hs, n), reps)
            out.append(f"{a:.3f} -> {c:.3f} (+{c-a:.3f}, {c/a:.3f}x{'' if (ra == rc).all() else ' MISMATCH'})")
        print(f"{name} 2^{lg}: resident blocking -> host scalars, ms: " + " | ".join(out), flush=True)
        ctx.bases_unregister(db.ptr)
        del db, ds

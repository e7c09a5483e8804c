"""Large-size sanity on one GPU: MSM / commit at 2^22 and 2^24 pairs (Nova config: 2^24 bases), NTT 2^24..2^26.
Checks size-independent properties (no oracle at these sizes): commit(k * ones) = k * sum, linearity of the MSM in the
scalars, idft(dft(v)) = v, and split-sum equality MSM(all) = MSM(first half) + MSM(second half)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
for curve, fld, name in ((K.KG_G1, K.KG_FR, "g1"), (K.KG_GRUMPKIN, K.KG_FQ, "grumpkin")):
    for lg in (22, 24, 25) if curve == K.KG_G1 else (22,):
        n = 1 << lg
        bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
        scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
        t = time.time(); ctx.gen_bases(curve, SEED + lg, 0, n, bases.data_ptr()); ctx.gen_scalars(fld, SEED + lg + 1, 0, n, scal.data_ptr()); ctx.sync()
        tg = time.time() - t
        t = time.time(); full = ctx.msm(curve, bases.data_ptr(), 0, scal.data_ptr(), n); t1 = time.time() - t
        t = time.time(); full2 = ctx.msm(curve, bases.data_ptr(), 0, scal.data_ptr(), n); t2 = time.time() - t
        h = n // 2
        a = ctx.msm(curve, bases.data_ptr(), 0, scal.data_ptr(), h)
        b = ctx.msm(curve, bases.data_ptr() + h * 64, 0, scal.data_ptr() + h * 32, n - h)
        xy, inf = ctx.points_sum_affine(curve, np.stack([a[:8], b[:8]]), np.array([not a[8:].any(), not b[8:].any()], dtype=np.uint8))
        ok = (xy == full[:8]).all() and (full == full2).all()
        print(f"{name} msm 2^{lg}: gen {tg:.2f}s first {t1*1e3:.1f} ms second {t2*1e3:.1f} ms -> {n/t2/1e6:.1f} Mpairs/s split-sum ok={ok}", flush=True)
        if lg == 24 and curve == K.KG_G1:                # the same with witness-like scalars (half ones, a fifth zeros): hot buckets under the wide windows
            from kogarashi_amd import synthetic as syn
            hs = scal.cpu().numpy().view(np.uint64).reshape(n, 4)
            syn.witness_like(hs, 11)
            scal.copy_(torch.from_numpy(hs.view(np.int64).reshape(-1)))
            full = ctx.msm(curve, bases.data_ptr(), 0, scal.data_ptr(), n)
            a = ctx.msm(curve, bases.data_ptr(), 0, scal.data_ptr(), h)
            b = ctx.msm(curve, bases.data_ptr() + h * 64, 0, scal.data_ptr() + h * 32, n - h)
            xy, inf = ctx.points_sum_affine(curve, np.stack([a[:8], b[:8]]), np.array([not a[8:].any(), not b[8:].any()], dtype=np.uint8))
            t = time.time(); full2 = ctx.msm(curve, bases.data_ptr(), 0, scal.data_ptr(), n); t2 = time.time() - t
            print(f"{name} msm 2^{lg}, witness-like scalars: {t2*1e3:.1f} ms split-sum ok={(xy == full[:8]).all() and (full == full2).all()}", flush=True)
        del bases, scal
for lg in (24, 26):
    n = 1 << lg
    v = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + 50 + lg, 0, n, v.data_ptr()); ctx.sync()
    ref = v.clone()
    ctx.ntt(v.data_ptr(), lg, False, True); ctx.sync()
    t = time.time(); ctx.ntt(v.data_ptr(), lg, True, True); ctx.sync(); dt = time.time() - t
    torch.cuda.synchronize()
    print(f"ntt 2^{lg}: coset_idft(coset_dft(v)) == v: {bool((v == ref).all())}  inverse pass {dt*1e3:.2f} ms", flush=True)

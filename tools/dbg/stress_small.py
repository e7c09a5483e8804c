"""Randomised stress of the short-input MSM (csrc/msm_small.hip) against the long pipeline on the same device arrays (the long pipeline is
what tests/test_gpu_parity.py and test_gpu_large.py hold against the oracle): random lengths 1 .. 32768 (G2 .. 20480), curves, scalar
patterns, identity flags, forced shapes (c, r) and the automatic one, blocking and in flight.  Exit code 1 on any mismatch.
   python tools/dbg/stress_small.py [rounds] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import kogarashi_amd as K
from kogarashi_amd import synthetic as syn
K.init()
ctx = K.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = torch.device("cuda", 0)
NMAX = 32768
store = {}
for curve in (0, 1, 2):
    fld = 1 if curve == 1 else 0
    s = torch.empty(NMAX * 4, dtype=torch.int64, device=dev)
    if curve == 2:
        b = torch.empty(NMAX * 16, dtype=torch.int64, device=dev)
        k = torch.empty(NMAX * 4, dtype=torch.int64, device=dev)
        ctx.gen_scalars(0, 31, 0, NMAX, k.data_ptr())
        inf = torch.zeros(NMAX, dtype=torch.uint8, device=dev)
        ctx.fixed_base_mul(2, k.data_ptr(), NMAX, b.data_ptr(), inf.data_ptr())
    else:
        b = torch.empty(NMAX * 8, dtype=torch.int64, device=dev)
        ctx.gen_bases(curve, 30 + curve, 0, NMAX, b.data_ptr())
        inf = torch.zeros(NMAX, dtype=torch.uint8, device=dev)
    store[curve] = (b, inf, s, fld)
ctx.sync()
bad = 0
for it in range(rounds):
    curve = int(rng.choice([0, 0, 1, 2]))
    b, inf, s, fld = store[curve]
    top = 20480 if curve == 2 else NMAX
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(1, 65)) if kind == 0 else int(rng.integers(1, 2200)) if kind == 1 else int(rng.integers(2000, 9000)) if kind == 2 else int(rng.integers(8000, top + 1))
    if rng.random() < 0.1:
        n = int(rng.choice([1536, 1537, 2048, 2049, 4096, 4097, 6144, 6145, 8192, 8193, 16384, top]))
    off = int(rng.integers(0, NMAX - n + 1))                  # a window of the stored bases
    W = 16 if curve == 2 else 8
    scal = np.empty((n, 4), dtype=np.uint64)
    ds = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(fld, 1000 + it, 0, n, ds.data_ptr()); ctx.sync()
    scal[:] = ds.cpu().numpy().view(np.uint64).reshape(n, 4)
    pat = int(rng.integers(0, 8))
    if pat == 0: scal[:] = scal[0]
    elif pat == 1: scal[rng.random(n) < 0.9] = scal[min(1, n - 1)]
    elif pat == 2: scal[:] = scal[:5][rng.integers(0, min(5, n), n)]
    elif pat == 3 and fld == 0: syn.witness_like(scal, it)
    elif pat == 4: scal[rng.random(n) < 0.5] = 0
    elif pat == 5: scal[rng.random(n) < 0.97] = scal[min(2, n - 1)]
    elif pat == 6: scal[:, 1:] = 0; scal[:, 0] &= np.uint64(0xFFFFF)          # Montgomery form of something: any 256-bit pattern below the modulus is a valid input
    ds.copy_(torch.from_numpy(scal.view(np.int64).reshape(-1)))
    flags = np.zeros(n, dtype=np.uint8)
    if rng.random() < 0.5:
        flags[rng.random(n) < rng.choice([0.01, 0.3, 1.0])] = 1
    dinf = torch.from_numpy(flags).to(dev)
    if curve == 2:
        dinf |= inf[off:off + n]
    bp = b.data_ptr() + off * W * 8
    ip = dinf.data_ptr() if (curve == 2 or flags.any() or rng.random() < 0.3) else 0
    ctx.set_msm_small(0)
    want = ctx.msm(curve, bp, ip, ds.data_ptr(), n)
    shapes = [(0, -1)]
    for _ in range(2):
        c = int(rng.integers(2, 11)); shapes.append((c, int(rng.integers(0, min(c - 1, 7) + 1))))
    for c, r in shapes:
        ctx.set_msm_small(NMAX, c, r)
        got = ctx.msm(curve, bp, ip, ds.data_ptr(), n)
        if not (got == want).all():
            bad += 1; print(f"MISMATCH it={it} curve={curve} n={n} off={off} pat={pat} shape=({c},{r}) blocking", flush=True)
    ctx.set_msm_small(NMAX, 0, -1)
    if n <= 8192:
        for t in range(3): ctx.msm_begin(curve, bp, ip, ds.data_ptr(), n, t)
        for t in range(3):
            if not (ctx.msm_end(curve, t) == want).all():
                bad += 1; print(f"MISMATCH it={it} curve={curve} n={n} in flight ticket {t}", flush=True)
    if it % 50 == 49: print(f"{it + 1} rounds, {bad} mismatches", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

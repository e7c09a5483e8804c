"""Groth16 2^18 throughput with k contexts on ONE GPU, each with its own resident CRS (window tables) and driven by its own
host thread, two proofs in flight per context:  python tools/dbg/g16_multi_ctx.py 1 2 3 [tables 0/1]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if os.environ.get('KG_WITH_TORCH') == '1':
    import torch
    torch.cuda.set_device(0)
    _t = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize()
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
from kogarashi_amd import synthetic as syn
from kogarashi_amd.api import groth16_setup
if os.environ.get('KG_HIPFLAGS'):
    import ctypes
    _hip = ctypes.CDLL('libamdhip64.so')
    print('hipSetDeviceFlags ->', _hip.hipSetDeviceFlags(int(os.environ['KG_HIPFLAGS'])))
if os.environ.get('KG_DUMMY_ALLOC'):
    import ctypes
    _hip = ctypes.CDLL('libamdhip64.so')
    _p = ctypes.c_void_p()
    print('hipMalloc ->', _hip.hipMalloc(ctypes.byref(_p), ctypes.c_size_t(int(os.environ['KG_DUMMY_ALLOC']) << 20)))
counts = [int(a) for a in sys.argv[1:] if a not in ("t0", "t1")] or [1, 2]
tables = "t0" not in sys.argv
m = 1 << 18
cc = syn.ChainCircuit(m)
c0 = K.Context(0)
P = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=c0)
r, s = syn.fixed_rs()
c0.close()
N = int(os.environ.get("KG_N", "48"))
for k in counts:
    ctxs = [K.Context(0) for _ in range(k)]
    state = []
    for c in ctxs:
        c.set_inputs_complete(True)
        pr = K.Prover(P, m, cc.l, cc.m_l_1, ctx=c, window_tables=tables)
        up = lambda v, c=c: c.upload(np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4))
        d = [up(v) for v in (cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w)]
        state.append((c, pr, d, (pr.crs, *[x.ptr for x in d], r, s)))
    results = [None] * k
    def worker(j, n):
        c, pr, d, args = state[j]
        c.groth16_prove_begin(*args, 0)
        last = None
        for i in range(1, n):
            c.groth16_prove_begin(*args, i & 1)
            last = c.groth16_prove_end((i - 1) & 1)
        results[j] = c.groth16_prove_end((n - 1) & 1)
    for j in range(k):
        worker(j, int(os.environ.get("KG_WARM", "24")))                     # warm-up (clock ramp of a cold GPU takes ~70 ms), one context at a time
    for c in ctxs:
        c.sync()
    ts = [threading.Thread(target=worker, args=(j, N)) for j in range(k)]
    t0 = time.perf_counter()
    if k == 1 and os.environ.get('KG_MAIN_THREAD') == '1':
        worker(0, N)
    else:
        for t in ts: t.start()
        for t in ts: t.join()
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    same = all(all((results[j][i] == results[0][i]).all() for i in range(4)) for j in range(k))
    print(f"{k} context(s), tables={tables}: {dt / (N * k) * 1e3:.3f} ms per proof ({N * k / dt:.0f} proofs/s), identical proofs: {same}", flush=True)
    del state
    for c in ctxs:
        c.close()

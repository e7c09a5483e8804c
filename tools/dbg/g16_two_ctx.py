"""Groth16 2^18 throughput with two contexts (own streams and work space each) driven by two host threads."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
captured = {}
orig = ctx.groth16_prove
def cap(*a):
    captured["args"] = a
    return orig(*a)
ctx.groth16_prove = cap
out = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), 18, steps=6, cpu=False)
print("one ctx: pipelined", round(out["ms_per_proof"], 3), "blocking", round(out["ms_per_proof_blocking"], 3))
args = captured["args"]
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctxs = [ctx] + [K.Context(0) for _ in range(nthreads - 1)]
N = 12
def worker(c, pipelined):
    if not pipelined:
        for _ in range(N):
            c.groth16_prove(*args) if c is not ctx else orig(*args)
    else:
        c.groth16_prove_begin(*args, 0)
        for i in range(1, N):
            c.groth16_prove_begin(*args, i & 1)
            c.groth16_prove_end((i - 1) & 1)
        c.groth16_prove_end((N - 1) & 1)
for pipelined in (False, True):
    for c in ctxs:
        (c.groth16_prove if c is not ctx else orig)(*args)
    torch.cuda.synchronize()
    ts = [threading.Thread(target=worker, args=(c, pipelined)) for c in ctxs]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nthreads} contexts, pipelined={pipelined}: {dt / (N * nthreads) * 1e3:.3f} ms per proof")

"""Host-side trace (KG_TRACE_HOST=1) of one blocking Groth16 proof: KG_TRACE_HOST=1 python tools/dbg/g16_host_trace.py log_m [call]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
ctx.set_inputs_complete(True)
calls = [0]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
orig = ctx.groth16_prove
def wrapped(*a, **k):
    calls[0] += 1
    mark = calls[0] == nth
    if mark: sys.stderr.write("[host] ---- proof begin\n"); sys.stderr.flush()
    t = time.perf_counter()
    out = orig(*a, **k)
    if mark: sys.stderr.write(f"[host] ---- proof end {(time.perf_counter() - t) * 1e6:.1f} us\n"); sys.stderr.flush()
    return out
ctx.groth16_prove = wrapped
o = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), int(sys.argv[1]), steps=4, cpu=False, from_witness=False)
print(o["ms_per_proof_blocking"])

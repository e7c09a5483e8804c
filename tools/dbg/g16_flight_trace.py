"""Proofs in flight only (for rocprofv3 --kernel-trace): python tools/dbg/g16_flight_trace.py log_m [count] ; BLOCKING=1 -> blocking calls"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
if os.environ.get("KG_ORDERED") != "1":
    ctx.set_inputs_complete(True)      # like bench.py: the inputs are uploaded and synchronised before the timed region
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
lg = int(sys.argv[1]); cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 12
grab = {}
orig = ctx.groth16_prove_begin
def first(*a):
    grab["args"] = a[:-1]
    return orig(*a)
ctx.groth16_prove_begin = first
bench.WARM_PROOFS = 2
bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), lg, steps=2, cpu=False, from_witness=False, tables=False) if "tables" in bench.bench_groth16.__code__.co_varnames else bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), lg, steps=2, cpu=False, from_witness=False)
ctx.groth16_prove_begin = orig
a = grab["args"]
ctx.sync(); time.sleep(0.01)
t0 = time.perf_counter()
if os.environ.get("BLOCKING"):
    for i in range(cnt): ctx.groth16_prove(*a)
else:
    for i in range(cnt):
        ctx.groth16_prove_begin(*a, i % 2)
        if i >= 1: ctx.groth16_prove_end((i - 1) % 2)
    ctx.groth16_prove_end((cnt - 1) % 2)
print(f"2^{lg}: {(time.perf_counter() - t0) / cnt * 1e3:.3f} ms per proof")

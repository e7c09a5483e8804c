"""GPU-side phase timeline of one Groth16 proof from the library's own HIP events (no profiler attached):
KG_PROFILE_TIMELINE=1 python tools/dbg/g16_timeline.py [log_m] [pipelined]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
if os.environ.get("KG_ORDERED") != "1":
    ctx.set_inputs_complete(True)
pipelined = len(sys.argv) > 2
calls = [0]
nth = int(os.environ.get("KG_TL_CALL", "3"))     # blocking mode: which call to record (3: plain CRS; 8: with window tables)
if pipelined:
    orig = ctx.groth16_prove_begin
    def wrapped(*a, **k):
        calls[0] += 1
        if calls[0] == 12:
            ctx.sync(); ctx.profile_enable(True)
        out = orig(*a, **k)
        if calls[0] == 15:
            ctx.profile_summary(); ctx.profile_enable(False)
        return out
    ctx.groth16_prove_begin = wrapped
else:
    orig = ctx.groth16_prove
    def wrapped(*a, **k):
        calls[0] += 1
        if calls[0] == nth:
            ctx.profile_enable(True)
        out = orig(*a, **k)
        if calls[0] == nth:
            ctx.profile_summary(); ctx.profile_enable(False)
        return out
    ctx.groth16_prove = wrapped
o = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), int(sys.argv[1]) if len(sys.argv) > 1 else 18, steps=4, cpu=False,
                        circuit=os.environ.get("KG_TL_CIRCUIT", "chain"), from_witness=False)
print(o["ms_per_proof"], o["ms_per_proof_blocking"])

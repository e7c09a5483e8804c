"""Groth16 2^18 proof time against a forced MSM window width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
for c in [int(a) for a in sys.argv[2:]]:
    ctx.set_msm_window(c)
    out = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), int(sys.argv[1]), steps=10, cpu=False)
    print(c, "pipelined", round(out["ms_per_proof"], 3), "blocking", round(out.get("ms_per_proof_blocking", 0), 3), flush=True)

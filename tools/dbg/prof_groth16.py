"""Runs only the Groth16 2^18 leg of bench.py (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = K.Context(0)
ctx.set_inputs_complete(True)      # as bench.py: the library's own queues, inputs synchronised before use
out = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), int(sys.argv[1]) if len(sys.argv) > 1 else 18, steps=5, cpu=False,
                          tickets=int(sys.argv[2]) if len(sys.argv) > 2 else 2)
print(out)

"""Soak of the prover's whole life cycle: setup -> CRS registered (+ window tables) -> proofs blocking / in flight / from a witness -> unregistered,
over and over on one context (bench.bench_groth16 at 2^14 and 2^16 constraints), watching memory.  usage: soak_g16.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
def rss_mb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"): return int(ln.split()[1]) / 1024
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hist = []
for r in range(rounds):
    for lg in (14, 17):
        out = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), lg, steps=6, cpu=False)
        assert out["pipelined_matches_blocking"] and out.get("window_tables", {}).get("proofs_match", True)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    fr, tot = ctx.mem_info()
    hist.append((round(rss_mb()), round((tot - fr) / 2**20)))
    print(f"round {r}: host RSS {hist[-1][0]} MiB, device in use {hist[-1][1]} MiB, 2^17 proof {out['ms_per_proof']:.3f} ms", flush=True)
base = hist[2]
assert hist[-1][0] - base[0] < 256 and hist[-1][1] - base[1] < 512, ("memory grew", base, hist[-1])
print("soak ok", base, hist[-1])

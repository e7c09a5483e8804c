"""Groth16 proof time against the constraint count (chain circuit, device setup): python tools/dbg/g16_sizes.py 8 10 12 ..."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()
gc.disable()         # no cyclic collection inside a timed loop (a 35 ms pause: tools/dbg/anom_1024.py)
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
if os.environ.get("KG_ORDERED") != "1":
    ctx.set_inputs_complete(True)      # like bench.py: the inputs are uploaded and synchronised before the timed region
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
for lg in [int(a) for a in sys.argv[1:]]:
    out = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), lg, steps=10, cpu=False, from_witness=False)
    wt = out.get("window_tables", {})
    print(f"m = 2^{lg}: in flight {out['ms_per_proof']:.3f} ms  blocking {out.get('ms_per_proof_blocking', 0):.3f}   tables: {wt.get('ms_per_proof', 0):.3f} / {wt.get('ms_per_proof_blocking', 0):.3f}"
          f"   us per constraint {out['ms_per_proof'] * 1e3 / (1 << lg):.3f}", flush=True)

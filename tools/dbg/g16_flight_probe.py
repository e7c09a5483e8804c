"""Host time inside kg_groth16_prove_begin / _end with two proofs in flight, against the blocking call: python tools/dbg/g16_flight_probe.py 10 12"""
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
acc = {}
def wrap(name):
    f = getattr(ctx, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
        acc.setdefault(name, []).append(dt * 1e6)
        return r
    setattr(ctx, name, g)
for nm in ("groth16_prove_begin", "groth16_prove_end", "groth16_prove"):
    wrap(nm)
for lg in [int(a) for a in sys.argv[1:]]:
    acc.clear()
    out = bench.bench_groth16(ctx, torch, dev, K, bench.single_rank_env(torch, dev), lg, steps=10, cpu=False, from_witness=False)
    med = lambda v: sorted(v)[len(v) // 2]
    n = 20
    print(f"m = 2^{lg}: in flight {out['ms_per_proof']:.3f} ms  blocking {out['ms_per_proof_blocking']:.3f} ms   begin {med(acc['groth16_prove_begin'][-n:]):.0f} us  end {med(acc['groth16_prove_end'][-n:]):.0f} us   prove {med(acc['groth16_prove'][-10:]):.0f} us", flush=True)
    print("   begin:", " ".join(f"{v:.0f}" for v in acc['groth16_prove_begin'][-12:]))
    print("   end:  ", " ".join(f"{v:.0f}" for v in acc['groth16_prove_end'][-12:]))

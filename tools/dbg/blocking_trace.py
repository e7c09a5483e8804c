"""A few blocking 2^20 G1 MSMs (for rocprofv3 --kernel-trace): python tools/dbg/blocking_trace.py [log_n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, kogarashi_amd as K
K.init()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = K.Context(0); ctx.set_inputs_complete(True)
dev = torch.device("cuda", 0)
n = 1 << lg
b = torch.empty(n * 8, dtype=torch.int64, device=dev); s = torch.empty(n * 4, dtype=torch.int64, device=dev)
ctx.gen_bases(0, 1, 0, n, b.data_ptr()); ctx.gen_scalars(0, 2, 0, n, s.data_ptr()); ctx.sync()
for _ in range(10): ctx.msm(0, b.data_ptr(), 0, s.data_ptr(), n)
if not os.environ.get('NOPROF'): ctx.profile_enable(True)
t0 = time.perf_counter()
for _ in range(reps): ctx.msm(0, b.data_ptr(), 0, s.data_ptr(), n)
dt = (time.perf_counter() - t0) / reps * 1e3
summ = ctx.profile_summary() if not os.environ.get('NOPROF') else {}
print(f"blocking 2^{lg}: {dt:.3f} ms  " + "  ".join(f"{k} {v[0] / v[1] * 1e3:.0f}us x{v[1] / reps:.1f}" for k, v in summ.items()))

"""Forward NTT time against log n:  python tools/dbg/ntt_sizes.py 14 16 18 20 22
KG_NTT_WARM=<ms> (default 60): untimed transforms for that long first -- an idle GPU needs tens of milliseconds of load to reach its
clock (2^22: 0.47 ms right after idle, 0.39 ms sustained on one box); KG_NTT_WARM=0 gives the cold figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
for lg in [int(a) for a in sys.argv[1:]]:
    n = 1 << lg
    v = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + lg, 0, n, v.data_ptr())
    for _ in range(3): ctx.ntt(v.data_ptr(), lg, False, False)
    torch.cuda.synchronize()
    t0 = time.time()
    while (time.time() - t0) * 1e3 < float(os.environ.get("KG_NTT_WARM", "60")):
        for _ in range(10): ctx.ntt(v.data_ptr(), lg, False, False)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ctx.ntt(v.data_ptr(), lg, False, False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"ntt 2^{lg}: {ms*1e3:.1f} us  {n/ms/1e6:.2f} G elements/s", flush=True)

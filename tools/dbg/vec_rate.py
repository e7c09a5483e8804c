"""Point-wise Fr vector ops (poly.rs:168-195): achieved bandwidth at 2^22 elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP
SEED = 0x4B6F676172617368
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = K.Context(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
n = 1 << 22
a = torch.empty(n * 4, dtype=torch.int64, device=dev); b = torch.empty_like(a); o = torch.empty_like(a)
ctx.gen_scalars(K.KG_FR, SEED, 0, n, a.data_ptr()); ctx.gen_scalars(K.KG_FR, SEED + 1, 0, n, b.data_ptr())
for op in ("add", "sub", "mul"):
    for _ in range(3): ctx.field_vec_op(K.KG_FR, op, a.data_ptr(), b.data_ptr(), o.data_ptr(), n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ctx.field_vec_op(K.KG_FR, op, a.data_ptr(), b.data_ptr(), o.data_ptr(), n)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{op}: {ms*1e3:.1f} us, {3*32*n/ms/1e6:.0f} GB/s")

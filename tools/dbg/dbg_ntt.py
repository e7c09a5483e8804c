import sys, time; sys.path.insert(0,'.')
import numpy as np, kogarashi_amd as K
from oracle import oracle as O
ctx=K.Context(0)
for k in (19,20,22):
    n=1<<k
    v=O.gen_scalars(0,123+k,0,n)
    d=ctx.upload(v)
    ctx.ntt(d.ptr,k,False,False); ctx.sync()
    t=time.time()
    for _ in range(5): ctx.ntt(d.ptr,k,False,False)
    ctx.sync(); dt=(time.time()-t)/5
    d2=ctx.upload(v); ctx.ntt(d2.ptr,k,False,False); got=d2.numpy()
    t=time.time(); want=O.Fft(k).dft(v,threads=16); to=time.time()-t
    print(k, "match", (got==want).all(), "gpu ms", dt*1e3, "oracle s", to, flush=True)

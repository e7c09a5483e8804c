"""G1 MSM step time with and without window tables (kg_bases_precompute), pipelined four deep and blocking:
python tools/dbg/table_rate.py 18 20        (KG_MERGED_T=<n> overrides the task length of the merged sort)"""
import sys, time
import numpy as np
import kogarashi_amd as K
K.init()          # one hardware queue per library queue (kg_init), before anything initialises HIP

ctx = K.Context(0)
ctx.set_inputs_complete(True)
for lg in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << lg
    db, ds = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, 11, 0, n, db.ptr)
    ctx.gen_scalars(K.KG_FR, 12, 0, n, ds.ptr)
    ctx.sync()
    ctx.bases_register(K.KG_G1, db.ptr, 0, n)

    def pipelined(k, depth=4):
        res = None
        for i in range(k):
            ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, i % 4)
            if i >= depth - 1:
                res = ctx.msm_end(K.KG_G1, (i - depth + 1) % 4)
        for i in range(max(k - depth + 1, 0), k):
            res = ctx.msm_end(K.KG_G1, i % 4)
        return res

    def measure(tag):
        pipelined(30)
        ctx.sync()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        r = pipelined(40)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 40 * 1e3
        ph = {k: round(v[0] / v[1], 3) for k, v in ctx.profile_summary().items()}
        ctx.profile_enable(False)
        t0 = time.perf_counter()
        for _ in range(10):
            rb = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
        bl = (time.perf_counter() - t0) / 10 * 1e3
        ctx.profile_enable(True)
        for _ in range(5):
            ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
        iso = {k: round(v[0] / v[1], 3) for k, v in ctx.profile_summary().items()}
        ctx.profile_enable(False)
        print(f"2^{lg} {tag:8s} pipelined {dt:.3f} ms/step  blocking {bl:.3f} ms  phases {ph}\n            isolated {iso}", flush=True)
        assert (r == rb).all()
        return r

    a = measure("plain")
    t0 = time.perf_counter()
    ctx.bases_precompute(db.ptr)
    ctx.sync()
    print(f"2^{lg} table build {(time.perf_counter() - t0) * 1e3:.1f} ms")
    b = measure("tables")
    assert (a == b).all(), "tables changed the result"
    ctx.bases_unregister(db.ptr)
ctx.close()

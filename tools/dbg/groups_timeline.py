"""Phase timeline (library HIP events) of one blocking kg_msm under the current KG_MSM_GROUPS:  python tools/dbg/groups_timeline.py lg [reg] [skew]
(reg: the bases registered first, as a resident commitment key or CRS is; skew: witness-like scalars -- half ones, a fifth zeros)"""
import os, sys
os.environ["KG_PROFILE_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kogarashi_amd as K
K.init()
ctx = K.Context(0)
lg = int(sys.argv[1]); n = 1 << lg
b, s = ctx.empty((n, 8)), ctx.empty((n, 4))
ctx.gen_bases(K.KG_G1, 1, 0, n, b.ptr); ctx.gen_scalars(K.KG_FR, 2, 0, n, s.ptr); ctx.sync()
if 'skew' in sys.argv[2:]:
    import numpy as np
    from kogarashi_amd import synthetic as syn
    hs = ctx.download(s); syn.witness_like(hs, 11); s = ctx.upload(hs)
if 'reg' in sys.argv[2:]: ctx.bases_register(K.KG_G1, b.ptr, 0, n)
ctx.set_inputs_complete(True)
for _ in range(5): ctx.msm(K.KG_G1, b.ptr, 0, s.ptr, n)
ctx.profile_enable(True)
ctx.msm(K.KG_G1, b.ptr, 0, s.ptr, n)
ctx.profile_summary(); ctx.profile_enable(False)

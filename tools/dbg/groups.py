"""Blocking kg_msm latency against the window-group plan (KG_MSM_GROUPS), one child process per setting:
   python tools/dbg/groups.py lg setting...      e.g.  groups.py 20 0 2 4 5,5,6"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import kogarashi_amd as K
    K.init()
    ctx = K.Context(0)
    lg = int(sys.argv[2]); n = 1 << lg
    curve = {"g1": K.KG_G1, "gk": K.KG_GRUMPKIN, "g2": K.KG_G2}[sys.argv[3]]
    reg = sys.argv[4] == "reg"
    b, s = ctx.empty((n, 16 if curve == K.KG_G2 else 8)), ctx.empty((n, 4))
    ctx.gen_bases(curve, 1, 0, n, b.ptr); ctx.gen_scalars(K.KG_FQ if curve == K.KG_GRUMPKIN else K.KG_FR, 2, 0, n, s.ptr); ctx.sync()
    if reg: ctx.bases_register(curve, b.ptr, 0, n)
    ctx.set_inputs_complete(True)
    for _ in range(5): r = ctx.msm(curve, b.ptr, 0, s.ptr, n)
    best = 1e9; tot = 0.0
    for _ in range(20):
        t = time.perf_counter(); r = ctx.msm(curve, b.ptr, 0, s.ptr, n); dt = time.perf_counter() - t
        best = min(best, dt); tot += dt
    ctx.profile_enable(True)
    for _ in range(5): ctx.msm(curve, b.ptr, 0, s.ptr, n)
    ph = ctx.profile_summary(); ctx.profile_enable(False)
    import hashlib
    print(f"groups={os.environ.get('KG_MSM_GROUPS', 'default'):10s} 2^{lg} {sys.argv[3]} {sys.argv[4]}: mean {tot / 20 * 1e3:.3f} ms  best {best * 1e3:.3f} ms  digest {hashlib.sha1(np.asarray(r).tobytes()).hexdigest()[:10]}  "
          + " ".join(f"{k}={v[0] / 5:.3f}" for k, v in ph.items()), flush=True)
    sys.exit(0)
lg = sys.argv[1]
curve = "g1"; reg = "plain"
settings = []
for a in sys.argv[2:]:
    if a in ("g1", "gk", "g2"): curve = a
    elif a in ("reg", "plain"): reg = a
    else: settings.append(a)
for rnd in range(2):
    for st in settings:
        env = dict(os.environ)
        if st == "default": env.pop("KG_MSM_GROUPS", None)
        else: env["KG_MSM_GROUPS"] = st
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", lg, curve, reg], env=env)

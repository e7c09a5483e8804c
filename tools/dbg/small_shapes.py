"""Latency of the short-input MSM (csrc/msm_small.hip) against its shape: python tools/dbg/small_shapes.py [curve] -> for every length the blocking
kg_msm time (median of 5 rounds of 8), the kernels' HIP-event time and the host finish, for every window width c and bucket range r."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import kogarashi_amd as K
K.init()
gc.disable()         # no cyclic collection inside a timed loop (a 35 ms pause: tools/dbg/anom_1024.py)
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
dev = torch.device("cuda", 0)
ctx = K.Context(0)
ctx.set_inputs_complete(True)
print(next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"), os.cpu_count(), "cpus")
nmax = max(lens)
s = torch.empty(nmax * 4, dtype=torch.int64, device=dev)
ctx.gen_scalars(1 if curve == 1 else 0, 77, 0, nmax, s.data_ptr())
if curve == 2:
    b = torch.empty(nmax * 16, dtype=torch.int64, device=dev)
    inf = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    ctx.fixed_base_mul(2, s.data_ptr(), nmax, b.data_ptr(), inf.data_ptr())
    ip = inf.data_ptr()
    ctx.gen_scalars(0, 78, 0, nmax, s.data_ptr())
else:
    b = torch.empty(nmax * 8, dtype=torch.int64, device=dev)
    ctx.gen_bases(curve, 76, 0, nmax, b.data_ptr())
    ip = 0
ctx.sync()


def lat(n, reps=8):
    f = lambda: ctx.msm(curve, b.data_ptr(), ip, s.data_ptr(), n)
    for _ in range(3):
        f()
    rr = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        rr.append((time.perf_counter() - t0) / reps * 1e3)
    ctx.profile_enable(True)
    for _ in range(4):
        f()
    summ = ctx.profile_summary()
    ctx.profile_enable(False)
    k_us = summ["small_msm"][0] / summ["small_msm"][1] * 1e3 if "small_msm" in summ else float("nan")
    lat.host_us = summ["host_finish"][0] / summ["host_finish"][1] * 1e3 if "host_finish" in summ else float("nan")
    return sorted(rr)[2], k_us


for n in lens:
    ctx.set_msm_small(0)
    base, _ = lat(n)
    ctx.set_msm_small(32768, 0, -1)
    auto, k_auto = lat(n)
    print(f"n = {n:5d}  long pipeline {base:.3f} ms   automatic shape {auto:.3f} ms (kernels {k_auto:.0f} us, host finish {lat.host_us:.0f} us)")
    row = []
    for c in range(2, 11):
        for r in range(0, min(c - 1, 7) + 1):
            if c - 1 - r > 5:
                continue
            try:
                ctx.set_msm_small(32768, c, r)
                ms, k_us = lat(n, reps=4)
            except Exception as e:
                continue
            row.append((ms, c, r, k_us, lat.host_us))
    row.sort()
    print("   best: " + "  ".join(f"c={c} r={r}: {ms:.3f} ({k_us:.0f} + {h_us:.0f} us)" for ms, c, r, k_us, h_us in row[:6]))
    print("   all : " + "  ".join(f"{c}/{r}:{ms:.3f}" for ms, c, r, k_us, h_us in sorted(row, key=lambda t: (t[1], t[2]))))

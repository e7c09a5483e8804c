#!/bin/bash
# A/B: unsplit blocking MSM with its reduction on the accumulation's queue (KG_BLOCKING_REDUCE_INLINE)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2 3; do
  for v in 0 1; do
    echo -n "$round KG_BLOCKING_REDUCE_INLINE=$v: "
    KG_BLOCKING_REDUCE_INLINE=$v python3 - <<'PY' 2>/dev/null
import time, torch, kogarashi_amd as K
K.init()
ctx = K.Context(0); ctx.set_inputs_complete(True)
dev = torch.device("cuda", 0)
out = []
for n in (8200, 1 << 14, 1 << 15, 1 << 16, 100000):
    b = torch.empty(n * 8, dtype=torch.int64, device=dev); s = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_bases(0, 1, 0, n, b.data_ptr()); ctx.gen_scalars(0, 2, 0, n, s.data_ptr()); ctx.sync()
    for _ in range(6): ctx.msm(0, b.data_ptr(), 0, s.data_ptr(), n)
    rr = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(8): r = ctx.msm(0, b.data_ptr(), 0, s.data_ptr(), n)
        rr.append((time.perf_counter() - t0) / 8 * 1e3)
    out.append(f"{n} {sorted(rr)[2]:.3f}")
    del b, s
print("  ".join(out))
PY
  done
done

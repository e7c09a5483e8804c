#!/bin/bash
# A/B of the lane-cooperative reduction tail (KG_COOP_TAIL): blocking kg_msm 2^16 .. 2^22, same box, alternating rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2 3; do
  for v in 0 1; do
    echo -n "$round KG_COOP_TAIL=$v: "
    KG_COOP_TAIL=$v python3 - <<'PY' 2>/dev/null
import time, torch, kogarashi_amd as K
K.init()
ctx = K.Context(0); ctx.set_inputs_complete(True)
dev = torch.device("cuda", 0)
out = []
for lg in (16, 18, 20, 22):
    n = 1 << lg
    b = torch.empty(n * 8, dtype=torch.int64, device=dev); s = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_bases(0, 1, 0, n, b.data_ptr()); ctx.gen_scalars(0, 2, 0, n, s.data_ptr()); ctx.sync()
    for _ in range(6): ctx.msm(0, b.data_ptr(), 0, s.data_ptr(), n)
    rr = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(6): r = ctx.msm(0, b.data_ptr(), 0, s.data_ptr(), n)
        rr.append((time.perf_counter() - t0) / 6 * 1e3)
    out.append(f"2^{lg} {sorted(rr)[2]:.3f} ms")
    del b, s
print("  ".join(out))
PY
  done
done

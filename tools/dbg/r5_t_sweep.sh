cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
timeout 900 python -m pytest tests/test_gpu_host_scalars.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
python tools/dbg/host_scalars.py 20 21 22 24 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5r/host.txt
python tools/dbg/msm_host_rate.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5r/host.txt
python -c "
import sys, time; sys.path.insert(0, '.')
import numpy as np, kogarashi_amd as K
from oracle import oracle as O
ctx = K.Context(0)
n = 1 << 10
b = O.gen_bases(0, 1, 0, n); s = O.gen_scalars(0, 2, 0, n); inf = np.zeros(n, dtype=np.uint8)
for _ in range(5): ctx.msm_host(K.KG_G1, b, inf, s, n)
t = time.perf_counter()
for _ in range(50): ctx.msm_host(K.KG_G1, b, inf, s, n)
print('kg_msm_host 2^10: %.3f ms' % ((time.perf_counter() - t) / 50 * 1e3))
" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5r/host.txt

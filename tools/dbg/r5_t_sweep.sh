cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
KG_STRESS_SEED=501 timeout 1200 python tools/dbg/stress_skew.py 60 2>&1 | grep -v amdgpu.ids | tail -4
KG_STRESS_SEED=502 KG_STRESS_LG=17,22 timeout 1500 python tools/dbg/stress_skew.py 40 2>&1 | grep -v amdgpu.ids | tail -4
KG_STRESS_SEED=503 KG_HOST_SLICES=5 KG_HOST_FIRST_DIV=3 timeout 1200 python tools/dbg/stress_skew.py 40 2>&1 | grep -v amdgpu.ids | tail -3

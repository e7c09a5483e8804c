cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
for lg in 14 15 16 17 18; do python tools/dbg/window_blocking.py $lg 11 12 13 14 15 16 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5s/window_blocking.txt

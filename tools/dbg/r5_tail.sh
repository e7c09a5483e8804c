cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
bash tools/dbg/ab_libs.sh python3 tools/dbg/host_timeline.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r5k/tail_ab.txt
cat gpurun_out/r5k/tail_ab.txt | cut -c1-400

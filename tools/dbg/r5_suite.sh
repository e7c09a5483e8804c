cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5h/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r5h/pytest_gpu.txt

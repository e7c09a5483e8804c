cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_nova.py tests/test_gpu_groth16.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4
python tools/dbg/setup_time.py 2>&1 | grep -v amdgpu.ids | tail -3

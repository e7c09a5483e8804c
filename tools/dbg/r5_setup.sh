cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
timeout 1500 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_nova.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -4
python tools/dbg/setup_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5p/setup_time.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5p/kt -- python3 tools/dbg/setup_time.py > /dev/null 2>&1
f=$(ls -t gpurun_out/r5p/kt/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/r5p/setup_kernel_stats.csv
timeout 900 python -m pytest tests/test_gpu_large.py -x -q -m gpu -k "groth16_2_18" 2>&1 | tail -3

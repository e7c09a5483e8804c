cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1500 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_parity.py::test_service_queues_are_placed_off_the_main_queues_pipe tests/test_gpu_groth16.py -x -q -m gpu > gpurun_out/r5i/pytest.txt 2>&1
tail -8 gpurun_out/r5i/pytest.txt
bash tools/dbg/r5_groups24.sh

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
g++ -O2 -std=c++17 -o /tmp/alloc_cost tools/host/alloc_cost.cpp -Lkogarashi_amd -lkogarashi_amd -Wl,-rpath,$PWD/kogarashi_amd || exit 1
g++ -O2 -std=c++17 -o /tmp/host_cost tools/host/host_cost.cpp -Lkogarashi_amd -lkogarashi_amd -Wl,-rpath,$PWD/kogarashi_amd || exit 1
/tmp/alloc_cost 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5m/alloc_cost.txt
KG_POOL_MB=0 /tmp/alloc_cost 2>&1 | grep -v amdgpu.ids | sed 's/^/KG_POOL_MB=0 /' | tee -a gpurun_out/r5m/alloc_cost.txt
/tmp/host_cost 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5m/host_cost.txt
/tmp/host_cost 22 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5m/host_cost.txt
KG_POOL_MB=0 /tmp/host_cost 20 2>&1 | grep -v amdgpu.ids | sed 's/^/KG_POOL_MB=0 /' | tee -a gpurun_out/r5m/host_cost.txt
timeout 1200 python -m pytest tests/test_gpu_holes.py tests/test_gpu_cpp_host.py tests/test_gpu_groth16.py -x -q -m gpu 2>&1 | tail -5

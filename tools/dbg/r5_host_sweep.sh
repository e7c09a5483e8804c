set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1200 python -m pytest tests/test_gpu_host_scalars.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r5b/pytest.txt 2>&1
tail -5 gpurun_out/r5b/pytest.txt
python tools/dbg/host_scalars.py 20 21 > gpurun_out/r5b/host_default.txt 2>&1
for kd in "2 2" "2 3" "3 2" "3 3" "4 2" "4 4"; do set -- $kd
  KG_HOST_SLICES=$1 KG_HOST_FIRST_DIV=$2 python tools/dbg/host_scalars.py 20 21 22 > gpurun_out/r5b/host_k$1_d$2.txt 2>&1
done
python tools/dbg/host_scalars.py 20 21 > gpurun_out/r5b/host_default2.txt 2>&1
for kd in "8 2" "8 3" "6 2"; do set -- $kd
  KG_HOST_SLICES=$1 KG_HOST_FIRST_DIV=$2 python tools/dbg/host_scalars.py 23 24 > gpurun_out/r5b/host24_k$1_d$2.txt 2>&1
done
tail -n 5 gpurun_out/r5b/host*.txt

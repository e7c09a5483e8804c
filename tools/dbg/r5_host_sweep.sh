cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
o=gpurun_out/r5d/sweep.txt; : > $o
run() { echo "== $*" >> $o; env "$@" python tools/dbg/host_scalars.py $SIZES 2>&1 | grep -v amdgpu.ids >> $o; }
SIZES="20"
run KG_HOST_ACCQ=1 KG_HOST_WINDOW_WHOLE=0
run KG_HOST_ACCQ=2 KG_HOST_WINDOW_WHOLE=0
run KG_HOST_ACCQ=1 KG_HOST_WINDOW_WHOLE=1
run KG_HOST_ACCQ=2 KG_HOST_WINDOW_WHOLE=1
for kd in "2 3" "2 4" "3 2" "3 4" "4 4"; do set -- $kd; run KG_HOST_SLICES=$1 KG_HOST_FIRST_DIV=$2; done
SIZES="21 22"
run KG_HOST_WINDOW_WHOLE=0
run KG_HOST_WINDOW_WHOLE=1
run KG_HOST_ACCQ=1
for kd in "2 2" "3 3" "4 2" "4 4"; do set -- $kd; run KG_HOST_SLICES=$1 KG_HOST_FIRST_DIV=$2; done
SIZES="23 24"
run KG_HOST_ACCQ=1
run KG_HOST_ACCQ=2
cat $o
timeout 900 python -m pytest tests/test_gpu_host_scalars.py -x -q -m gpu 2>&1 | tail -3

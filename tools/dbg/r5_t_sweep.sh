cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_gpu_host_scalars.py -x -q -m gpu 2>&1 | tail -8
o=gpurun_out/r5g/cuts.txt; : > $o
for c in "" 0.33 0.5 "0.1,0.4" "0.08,0.24,0.5" "0.15,0.5" "0.2,0.6"; do
  echo "== KG_HOST_CUTS=$c" >> $o
  KG_HOST_CUTS=$c KG_PROFILE_TIMELINE=1 python tools/dbg/host_timeline.py 20 2>&1 | grep -v "amdgpu.ids\|^\[host\]" | sed -n '1,/---- registered/p' >> $o
done
echo "== KG_HOST_SHARED=0" >> $o
KG_HOST_SHARED=0 python tools/dbg/host_scalars.py 20 21 22 23 2>&1 | grep -v amdgpu.ids >> $o
echo "== shared" >> $o
python tools/dbg/host_scalars.py 20 21 22 23 24 2>&1 | grep -v amdgpu.ids >> $o
grep "==\|round 1\|accumulate\|sort \|reduce\|gather\|g1 2" $o

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
o=gpurun_out/r5i/groups24.txt; : > $o
for rep in 1 2; do
for g in auto "1,4,4,4" "2,3,4,4" "1,3,4,5" "2,4,7" "1,5,7" "3,3,3,4"; do
  if [ $g = auto ]; then python tools/dbg/commit24.py 24 2>&1 | grep -v amdgpu.ids >> $o; else KG_MSM_GROUPS=$g python tools/dbg/commit24.py 24 2>&1 | grep -v amdgpu.ids >> $o; fi
done; done
for g in auto "1,4,4,4" "2,3,4,4"; do
  if [ $g = auto ]; then python tools/dbg/commit24.py 24 w 2>&1 | grep -v amdgpu.ids | sed 's/^/witness-like /' >> $o; else KG_MSM_GROUPS=$g python tools/dbg/commit24.py 24 w 2>&1 | grep -v amdgpu.ids | sed 's/^/witness-like /' >> $o; fi
done
cat $o

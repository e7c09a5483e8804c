cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
o=gpurun_out/r5q/gsnt.txt; : > $o
for rep in 1 2; do
for v in 0 512 1024; do
  echo "== KG_GS_NT=$v" >> $o
  KG_GS_NT=$v python tools/dbg/host_scalars.py 20 21 2>&1 | grep -v amdgpu.ids >> $o
done; done
cat $o

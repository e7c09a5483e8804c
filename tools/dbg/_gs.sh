mkdir -p gpurun_out/r4bg
run() { env "$@" python tools/dbg/groups.py $LG default reg 2>/dev/null | grep groups= | sed "s/^/$* /" | sed 's/groups=default *//; s/digest [0-9a-f]* //'; }
for LG in 24 20; do
run KG_GS_TILE=0
run KG_GS_TILE=8192
run KG_GS_TILE=8192 KG_GS_NT0=1024
run KG_GS_TILE=8192 KG_GS_NT0=1024 KG_GS_NT=512
run KG_GS_TILE=8192 KG_GS_NT=1024
run KG_GS_TILE=4096 KG_GS_NT0=1024
run KG_GS_TILE=4096 KG_GS_NT0=1024 KG_GS_NT=512
done > gpurun_out/r4bg/g.txt
bash tools/dbg/ab_knobs.sh gpurun_out/r4bg/ab 2 KG_GS_TILE=0 KG_GS_TILE=8192 "KG_GS_TILE=8192 KG_GS_NT=512" "KG_GS_TILE=4096 KG_GS_NT=512" "KG_GS_TILE=8192 KG_GS_NT=1024" > gpurun_out/r4bg/ab.txt 2>&1
cat gpurun_out/r4bg/g.txt gpurun_out/r4bg/ab.txt

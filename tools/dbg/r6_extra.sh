#!/bin/bash
# round 6, beyond tools/collect_profiles.sh: the ladders and the A/B records of the round's changes
#   gpurun --timeout 2400 -- 'bash tools/dbg/r6_extra.sh'     -> gpurun_out/r6_extra/*.txt (copied into profiles/r06_*.txt by hand)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_extra; rm -rf "$O"; mkdir -p "$O"
# 1. blocking and in-flight MSM over the ladder of lengths, three curves (the short-input kernel up to 32768 pairs, G2 20480)
timeout -s KILL 400 python3 tools/dbg/size_sweep.py 4 24 g1 > "$O/size_sweep.txt" 2>&1
timeout -s KILL 300 python3 tools/dbg/size_sweep.py 4 20 gk > "$O/size_sweep_grumpkin.txt" 2>&1
timeout -s KILL 300 python3 tools/dbg/size_sweep.py 4 18 g2 > "$O/size_sweep_g2.txt" 2>&1
# 2. A/B of the short-input kernel against the long pipeline (KG_SMALL_MAX=0), same box, alternating, two rounds
for round in 1 2; do
  for v in 0 32768; do
    echo "== round $round KG_SMALL_MAX=$v"
    KG_SMALL_MAX=$v timeout -s KILL 120 python3 tools/dbg/size_sweep.py 4 15 g1 2>/dev/null | grep -E "n = +(16|32|64|256|1024|4096|8192|16384|32768) "
  done
done > "$O/small_ab.txt" 2>&1
# 2b. A/B of the halved scalars (GLV, KG_SMALL_GLV) on blocking and in-flight MSMs, G1 and G2, two alternating rounds
for round in 1 2; do
  for v in 0 1; do
    echo "== round $round G1 KG_SMALL_GLV=$v"
    KG_SMALL_GLV=$v timeout -s KILL 120 python3 tools/dbg/size_sweep.py 4 13 g1 2>/dev/null | grep -E "n = +(16|32|64|256|1024|2048|4096|6144|8192) "
    echo "== round $round G2 KG_SMALL_GLV=$v"
    KG_SMALL_GLV=$v timeout -s KILL 120 python3 tools/dbg/size_sweep.py 4 14 g2 2>/dev/null | grep -E "n = +(16|32|64|256|1024|4096|8192|16384) "
  done
done > "$O/small_glv_ab.txt" 2>&1
# 3. every shape of the short-input kernel (window width x bucket range), three curves
timeout -s KILL 300 python3 tools/dbg/small_shapes.py 0 16,32,64,128,256,512,1024,2048,4096,8192,16384,32768 > "$O/small_shapes_g1.txt" 2>&1
timeout -s KILL 300 python3 tools/dbg/small_shapes.py 2 16,64,256,1024,4096,16384,32768 > "$O/small_shapes_g2.txt" 2>&1
timeout -s KILL 300 python3 tools/dbg/small_shapes.py 1 32,1024,4096,16384,32768 > "$O/small_shapes_grumpkin.txt" 2>&1
# 4. phase times inside the short-input kernel (A/B build with the stamps)
for a in "16 2 1" "256 2 1" "1024 2 1" "1024 4 1" "2048 4 1" "16384 8 3"; do echo "== n c r = $a"; timeout -s KILL 60 python3 tools/dbg/small_stamps.py $a 2>&1 | grep "^\[small\]" | tail -2; done > "$O/small_stamps.txt" 2>&1
# 5. the prover over its lengths (short proofs: five one-launch MSMs)
timeout -s KILL 600 python3 tools/dbg/g16_sizes.py 4 6 8 10 12 14 16 18 > "$O/g16_ladder.txt" 2>&1
for v in 0 32768; do echo "== KG_SMALL_MAX=$v"; KG_SMALL_MAX=$v timeout -s KILL 300 python3 tools/dbg/g16_sizes.py 6 10 12 13 14 2>/dev/null; done > "$O/g16_small_ab.txt" 2>&1
for v in 0 1; do echo "== KG_SMALL_GLV=$v"; KG_SMALL_GLV=$v timeout -s KILL 300 python3 tools/dbg/g16_sizes.py 4 6 8 10 12 2>/dev/null; done > "$O/g16_glv_ab.txt" 2>&1
echo "== KG_ORDERED=1 (the default mode of a context: inputs ordered behind the caller's queue)" > "$O/g16_ladder_ordered.txt"; KG_ORDERED=1 timeout -s KILL 300 python3 tools/dbg/g16_sizes.py 4 6 8 10 12 14 2>/dev/null >> "$O/g16_ladder_ordered.txt"
# 6. the lane-cooperative reduction tail against the streamed one (blocking kg_msm), three rounds
bash tools/dbg/r6_tail_ab.sh > "$O/tail_coop_ab.txt" 2>&1
ls -la "$O"

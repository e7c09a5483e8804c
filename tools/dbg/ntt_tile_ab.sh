#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/ab1; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for t in 0 11 12; do
  KG_NTT_TILE=$t timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${t}_$rep -- python3 tools/dbg/ntt_sizes.py 22 > $O/out_${t}_$rep.txt 2>$O/err_${t}_$rep.txt
  echo "tile=$t rep=$rep"; grep "^ntt" $O/out_${t}_$rep.txt; python3 tools/dbg/kstats.py $O/kt_${t}_$rep | grep ntt_tile
done; done

#!/bin/bash
# Times every build/exp/libkg_*.so (tools/dbg/build_variants.sh) on this box, two alternating rounds; KG_NTT_TILE / KG_NTT_STEPS
# pass through:   gpurun -- 'KG_NTT_TILE=11 bash tools/dbg/run_ntt_variants.sh 20 22'
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for so in build/exp/libkg_*.so; do
    n=${so##*/libkg_}; n=${n%.so}
    echo "$rep $n: $(KG_LIB_PATH=$PWD/$so timeout -s KILL 120 python3 tools/dbg/ntt_sizes.py "$@" 2>&1 | grep '^ntt' | sed 's/ntt 2^//; s/  .*//' | tr '\n' ' ')"
  done
done

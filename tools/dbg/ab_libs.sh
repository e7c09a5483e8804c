#!/bin/bash
# Runs one command under every build/exp/libkg_*.so (tools/dbg/build_variants.sh), three alternating rounds:
#   gpurun -- 'bash tools/dbg/ab_libs.sh python3 tools/dbg/window_pipe.py 20 16'
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for so in build/exp/libkg_*.so; do
    n=${so##*/libkg_}; n=${n%.so}
    echo "$rep $n: $(KG_LIB_PATH=$PWD/$so timeout -s KILL 300 "$@" 2>&1 | grep -v -i "amdgpu.ids" | tail -3 | tr '\n' ' ')"
  done
done

#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  echo "old: $(KG_LIB_PATH=$PWD/build/exp/libkg_base.so python3 tools/dbg/ntt_sizes.py 18 20 22 2>&1 | grep '^ntt' | sed 's/ntt 2^//; s/  .*//' | tr '\n' ' ')"
  echo "new: $(python3 tools/dbg/ntt_sizes.py 18 20 22 2>&1 | grep '^ntt' | sed 's/ntt 2^//; s/  .*//' | tr '\n' ' ')"
done

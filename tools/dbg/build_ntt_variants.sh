#!/bin/bash
# Phase-off builds of the transform for A/B timing on one box:  bash tools/dbg/build_ntt_variants.sh
# -> build/exp/libkg_<name>.so (same ABI; select with KG_LIB_PATH).  Variants: name=flags, flags joined by '+'.
set -eu
cd "$(dirname "$0")/../.."
python3 -m kogarashi_amd.build > /dev/null
mkdir -p build/exp
C=kogarashi_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -ffp-contract=off -Xarch_host -march=x86-64-v3 -w"
for v in "$@"; do
  name=${v%%=*}; defs=${v#*=}; D=""
  for d in ${defs//+/ }; do D="$D -D$d"; done
  ( hipcc -x hip $FL $D -c $C/ntt.hip -o build/exp/ntt_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libkg_$name.so $C/capi.o $C/sharded.o $C/vec.o $C/msm.o build/exp/ntt_$name.o $C/groth16.o && echo built $name ) &
done
wait

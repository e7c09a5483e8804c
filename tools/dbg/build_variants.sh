#!/bin/bash
# Experiment builds of one source for A/B timing on one box:
#   bash tools/dbg/build_variants.sh msm_run.hip base=KG_BASE tail512=KG_TAIL_L=512
#   bash tools/dbg/build_variants.sh ntt.hip base=KG_NTT_BASE nomem=KG_NTT_EXP_NOMEM all4=KG_NTT_EXP_NOBAR+KG_NTT_EXP_NOTW+...
# -> build/exp/libkg_<name>.so (same ABI; select with KG_LIB_PATH).  name=flags, flags joined by '+', each becomes -D<flag>.
set -eu
cd "$(dirname "$0")/../.."
python3 -m kogarashi_amd.build > /dev/null
mkdir -p build/exp
C=kogarashi_amd/csrc
SRC=$1; shift
STEM=${SRC%.*}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -ffp-contract=off -Xarch_host -march=x86-64-v3 -w"
for v in "$@"; do
  name=${v%%=*}; defs=${v#*=}; D=""
  for d in ${defs//+/ }; do D="$D -D$d"; done
  ( hipcc -x hip $FL $D -c $C/$SRC -o build/exp/${STEM}_$name.o
    OBJS=""
    for o in capi tuning sharded msm_host vec msm_sort msm_run msm_small ntt groth16 setup; do
      if [ $o = $STEM ]; then OBJS="$OBJS build/exp/${STEM}_$name.o"; else OBJS="$OBJS $C/$o.o"; fi
    done
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libkg_$name.so $OBJS && echo built $name ) &
done
wait

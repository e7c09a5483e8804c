#!/bin/bash
# uneven window groups of a blocking 2^20 / 2^19 / 2^21 MSM (KG_MSM_GROUPS lists the groups' window counts from the top window down)
for r in 1 2; do
for g in auto 8,8 6,10 5,11 4,12 10,6 4,6,6 3,5,8; do
  if [ $g = auto ]; then unset KG_MSM_GROUPS; else export KG_MSM_GROUPS=$g; fi
  echo "groups=$g: $(python tools/dbg/window_blocking.py 20 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
done
done
unset KG_MSM_GROUPS
for g in auto 9,8 6,11 5,12 4,13; do
  if [ $g = auto ]; then unset KG_MSM_GROUPS; else export KG_MSM_GROUPS=$g; fi
  echo "2^18 groups=$g: $(python tools/dbg/window_blocking.py 18 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
done

#!/bin/bash
mkdir -p gpurun_out/r4m
for r in 1 2 3; do
  KG_MSM_GROUPS=0 python tools/dbg/groups.py 20 0 2>&1 | grep -v amdgpu | head -1
  KG_GROUP_MAIN_FIRST=1 python tools/dbg/groups.py 20 2 2>&1 | grep -v amdgpu | head -1 | sed 's/^/mf1 /'
  KG_GROUP_MAIN_FIRST=0 python tools/dbg/groups.py 20 2 2>&1 | grep -v amdgpu | head -1 | sed 's/^/mf0 /'
  KG_GROUP_MAIN_FIRST=0 KG_GROUP_ACCQ=1 python tools/dbg/groups.py 20 2 2>&1 | grep -v amdgpu | head -1 | sed 's/^/mf0q1 /'
done > gpurun_out/r4m/ab.txt 2>&1

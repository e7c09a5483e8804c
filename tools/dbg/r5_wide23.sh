#!/bin/bash
for r in 1 2; do
python tools/dbg/commit24.py 23 2>&1 | grep -v amdgpu.ids
KG_WIDE_WINDOW=23 python tools/dbg/commit24.py 23 2>&1 | grep -v amdgpu.ids
KG_FORCE_C=19 python tools/dbg/commit24.py 23 2>&1 | grep -v amdgpu.ids
KG_FORCE_C=19 KG_MSM_GROUPS=3 python tools/dbg/commit24.py 23 2>&1 | grep -v amdgpu.ids
KG_WIDE_WINDOW=23 KG_MSM_GROUPS=3 python tools/dbg/commit24.py 23 2>&1 | grep -v amdgpu.ids
done
python tools/dbg/commit24.py 22 2>&1 | grep -v amdgpu.ids
KG_FORCE_C=19 python tools/dbg/commit24.py 22 2>&1 | grep -v amdgpu.ids
KG_FORCE_C=19 python tools/dbg/commit24.py 24 2>&1 | grep -v amdgpu.ids
python tools/dbg/commit24.py 24 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# host-scalar MSM at 2^16..2^19: slices 1 / 2, without and with window tables (TABLES=1 builds them before the timing)
for r in 1 2; do
for k in 1 2; do
  for tb in 0 1; do
    echo "== KG_HOST_SLICES=$k TABLES=$tb round $r"
    KG_HOST_SLICES=$k TABLES=$tb python tools/dbg/host_scalars.py 16 17 18 19 2>&1 | grep -v amdgpu.ids
  done
done
done

#!/bin/bash
for lg in 4 6 8 9 10 11 12 13; do python tools/dbg/window_blocking.py $lg 4 5 6 7 8 9 10 11 12 13 14 15 16 2>&1 | grep -v amdgpu.ids; done
